// lin_kernels.hip -- the train evaluation in the scaled linear semiring (rules: lin_rules.h), gfx950.
//
// Same diagonal-synchronous batch pipeline as train_kernels.hip (a group of G sequences swept in
// lockstep, kernel boundaries = dependencies), but a term of the O(L W^2) rules is one FMA on two
// table loads instead of an exp.  The band kernels are bound by instruction issue and by the dependent round trips of a
// workgroup (HBM traffic is at 1.3 x the algorithmic bytes and 40 % of the peak), hence table-driven phases,
// several blocks of cells per workgroup and as many resident workgroups as registers and LDS allow:
//   k4_weights   once per evaluation: exp(lambda_k * structural term) of every cell (the loop items' terms are
//                exponentiated where their records are staged)
//   k4_in(d)     ONE launch per diagonal: workgroup = nblk blocks of cpb = 256/lanes-per-cell consecutive cells of one sequence;
//                set-up once (context of all its cells staged in LDS), then per block: pair phase (rule 2 factorised: lane = (cell, pair)), item sums (rule 6c: lane =
//                (record, tuple), records staged in LDS), then one lane per (cell, state) finishes P,E,M,B,1,2,L -- the heavy
//                sums never touch HBM
//   k4_in_ext    exterior chain, partition functions, objective, range check (flags the sequence)
//   k4_out_ext   exterior chain of the outside pass;  k4_r7: its contribution to every pair cell (rule 7)
//   k4_out(d)    outside sweep with the expected counts; train schedule 1 = ONE sweep for both passes of the reference: the
//                "has motif" terminals on the pattern's states, the "no motif" terminal on a shadow copy of state (0,0),
//                each world with its own Z and statistics
//   k4_combine   statistics of the reference's two passes (motif_trainer.hpp:209-225) from those two
//   k5_*         scan: position arg-max, Viterbi pass (max-plus, reference tie order), traceback
// Reference path: RNAelemTrainDP::operator(), RNAelem/motif_trainer.hpp:124-272.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#ifndef ELEMDP_KCI
#define ELEMDP_KCI 2
#endif
#ifndef ELEMDP_LB_IN
#define ELEMDP_LB_IN 4
#endif
// The table-driven k4_in needs 65 vector registers.  Asked for eight waves per SIMD (W8) it fits 64 with a few spilled dwords:
// worth it where the eighth workgroup per CU then fits the LDS as well (113.4 vs 115.4 ms per 4096 x L=200 for ((.*.)): 19.7 KB
// per workgroup), a loss where the LDS holds fewer workgroups anyway ((.....): 24.4 KB, +1 %) -- the launcher chooses.
#ifndef ELEMDP_LB_IN_FAST
#define ELEMDP_LB_IN_FAST 8
#endif
// k5_cyk: 64 registers without spills when asked for eight waves per SIMD, and 17.9 KB of LDS with two split points per staging
// round (25.9 KB with four): eight workgroups per CU instead of six, scan 982 -> 961 ms (10 000 x L=300)
#ifndef ELEMDP_LB_CYK
#define ELEMDP_LB_CYK 8
#endif
#ifndef ELEMDP_LB_OUT6
#define ELEMDP_LB_OUT6 6
#endif
#ifndef ELEMDP_LB_OUT
#define ELEMDP_LB_OUT 4
#endif
// workgroup size of the band kernels (k4_in, k4_out, k5_cyk) and of the helpers they share
#ifndef ELEMDP_BAND_THREADS
#define ELEMDP_BAND_THREADS 256
#endif
#ifndef ELEMDP_CPB_MAX
#define ELEMDP_CPB_MAX 64
#endif
#ifndef ELEMDP_KIB
#define ELEMDP_KIB 4
#endif
#ifndef ELEMDP_RECIN
#define ELEMDP_RECIN 1024
#endif
#ifndef ELEMDP_RECOUT
#define ELEMDP_RECOUT 960
#endif
#include "kernels.h"
#include "lin_rules.h"
#include "lin_fast.h"
#include "scan_rules.h"
#include "wave_gather.h"

namespace elemdp {
namespace {
constexpr int kBT = ELEMDP_BAND_THREADS;   // threads of a band-kernel workgroup
constexpr int kWaves = kBT / 64;

// q / n for small non-negative q (< 2^20) and n > 0 through the single-precision reciprocal: (q + 1/2) * rcp(n) is off by less than
// 1 / (2 n) from the true quotient plus 1/(2n), so the truncation is exact.  An integer division by a run-time divisor costs ~30
// instructions, and the band kernels are bound by instruction issue.
__device__ __forceinline__ int div_small(int q, int n) { return (int)(((float)q + 0.5f) * __frcp_rn((float)n)); }
// ... with the reciprocal formed on the host (LinArgs::rcp_*: 1.0f / n, the same rounding)
__device__ __forceinline__ int div_rcp(int q, float rcp) { return (int)(((float)q + 0.5f) * rcp); }

// Deterministic mode (LinArgs::det, option "deterministic"): two evaluations of the same batch are bit-identical (the reference at
// --thread 1: motif_trainer.hpp:248-271).  An LDS sum that lanes of DIFFERENT waves add to has no fixed order; the adds of ONE wave
// reach LDS in program order, lanes of one instruction in lane order.  So every heavy sum of a workgroup gets its adds from one
// wave only: the pairs of a cell sit in one wave (LinArgs::det_sh: a power of two of lanes per cell), the tuples of the item sums
// are dealt to the waves by target (AutomatonLayout::qd_*), the stem cells of HA by wave; the expected counts, which every lane may
// add to, keep one copy per wave, added in wave order where they are flushed (rep_sum); a workgroup adds its counts to a row of
// its own (sequence, block) instead of the sequence's row, and k4_combine sums those rows in block order.  (Round 3 kept a copy
// of EVERY shared sum per wave: 32 KB of LDS for k4_out, half the resident workgroups, 1.7 - 1.9 x the time.)
__device__ __forceinline__ double rep_sum(const double* p, int nrep, int stride) {
  double a = p[0];
  for (int r = 1; r < nrep; ++r) a += p[r * stride];
  return a;
}

// statistics sink of the linear pipeline: emission counts into LDS, energy statistics lane-private
struct LinSink {
  double* en_;
  double eh0, eh1;
  // merged schedule: the lane works in world 1 (the shadow copy of (0,0) = the "no motif" pass) -- its statistics go to the
  // second set of accumulators
  int world = 0;
  double* pos0 = nullptr;   // scan: per-sequence position posteriors (global memory, linear): start, inner, end
  double* pos1 = nullptr;
  double* pos2 = nullptr;
  __device__ __forceinline__ void en(int idx, double w) { atomicAdd(&en_[idx], w); }
  // (selects, not branches: a branch per accumulator becomes a pointer to one of the fields, and the whole sink then lives
  // in scratch memory -- 96 B per lane, a memory round trip per statistic and 21 MB of HBM writes per sequence)
  __device__ __forceinline__ void eh(int k, double w) { eh1 += k ? w : 0.; eh0 += k ? 0. : w; }
  __device__ __forceinline__ void pos(int which, int p, double w) {
    double* a = (which == 0) ? pos0 : (which == 1) ? pos1 : pos2;
    if (a) atomicAdd(&a[p], w);
  }
};

// phase timer (option "profile"): thread 0 of a workgroup sums the shader clocks between marks per slot and adds
// them to one of 64 copies of the counter row when the workgroup ends (host adds the copies)
struct PhaseClock {
  // (the sums go straight to the counter row: accumulators in registers -- thirteen 64-bit values -- would stay live around the
  // whole block loop of a band kernel, profiled or not; the last clock is read by every lane, outside divergent control flow,
  // and so stays in scalar registers)
  long long* row;
  long long t0;
  __device__ __forceinline__ void start(long long* p) {
    row = p ? p + 16 * ((blockIdx.x + 7 * blockIdx.y) & 63) : nullptr;
    t0 = 0;
    if (row) t0 = __builtin_readcyclecounter();
  }
  template <int SLOT> __device__ __forceinline__ void mark() {
    if (row) {
      const long long t = __builtin_readcyclecounter();
      if (threadIdx.x == 0) atomicAdd((unsigned long long*)&row[SLOT], (unsigned long long)(t - t0));
      t0 = t;
    }
  }
  __device__ __forceinline__ void finish() {}
};

struct LViews {
  ModelView m;
  // `lay` must live in LDS (stage_layout): every field access of a layout in global memory is a vector load with its
  // own wait, dozens of serialized round trips per target
  __device__ explicit LViews(const AutomatonLayout& lay) : m(lay) {}
  SeqView q;
  TableView in, out;
  int n;
  bool positive;
  long long seq_base, pos_base;
  double* row;
  double* zs;
};

__device__ __forceinline__ void make_lviews(const LinArgs& a, int g, LViews& v) {
  // the plan of the slot: one uniform record (g is the same for the whole workgroup: scalar loads), no grp -> plans chain
  // (read through the constant address space: the record is not written while kernels run, and a uniform address then
  // becomes scalar loads -- one wait for the whole record)
  g = __builtin_amdgcn_readfirstlane(g);
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const SeqPlan __attribute__((address_space(4))) * ConstPlan;
  const SeqPlan p = *reinterpret_cast<ConstPlan>(reinterpret_cast<uintptr_t>(a.plans_slot + g));
#else
  const SeqPlan p = a.plans_slot[g];
#endif
  const int n = p.index;
  v.n = n;
  v.positive = p.positive != 0;
  v.seq_base = p.seq_base;
  v.pos_base = p.pos_base;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const ParamBlock __attribute__((address_space(4))) * ConstParams;   // (uniform, read-only: scalar loads)
  const ParamBlock pbv = *reinterpret_cast<ConstParams>(reinterpret_cast<uintptr_t>(a.params));
#else
  const ParamBlock pbv = *reinterpret_cast<const ParamBlock*>(a.params);
#endif
  const ParamBlock* pb = &pbv;
  v.m.ints = a.ints;
  v.m.big = a.ints;
  v.m.theta = a.params + sizeof(ParamBlock) / sizeof(double);
  v.m.lin = a.lin;
  v.m.lambda[0] = pb->lambda[0];
  v.m.lambda[1] = pb->lambda[1];
  v.m.log_tau = pb->log_tau;
  v.m.lam_same = pb->lam_same;
  v.m.no_prf = a.no_prf;
  v.m.m_min = a.m_min;
  v.m.dbg = a.dbg;
  SeqView& q = v.q;
  q.L = p.L; q.W = p.W; q.C = p.C;
  q.seq = a.b.seq + p.seq_base;
  q.ws = a.b.ws + p.pos_base;
  q.ews = a.ews + p.pos_base;
  q.unp = a.b.unp + p.pos_base;
  q.okbits = a.okbits + p.bits_base;
  q.dmin = a.p.dmin + p.dmin_base;
  q.e_stack = a.p.e_stack + p.cell_base; q.e_ext = a.p.e_ext + p.cell_base; q.e_ml = a.p.e_ml + p.cell_base;
  q.e_close = a.p.e_close + p.cell_base; q.e_hp = a.p.e_hp + p.cell_base;
  q.xwc = a.xwc + p.cell_base; q.xwc_stride = a.xwc_stride;
  q.xwi = a.xwi + p.item_base; q.xwi_stride = a.xwi_stride;
  q.items_inner = a.p.items_inner + p.item_base; q.items_left = a.p.items_left + p.item_base;
  q.items_right = a.p.items_right + p.item_base;
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
  v.in.ap = a.a_in ? a.a_in + (size_t)g * a.a_stride : nullptr;
  v.out.ap = a.a_out ? a.a_out + (size_t)g * a.a_stride : nullptr;
  v.in.nA = v.out.nA = a.lay.n_ap;
  v.in.set_compact(a.lay, a.ints);      // (the column map moves to LDS with the automaton blob: stage_context)
  v.out.set_compact(a.lay, a.ints);
  v.in.cyk_compact = a.cyk_compact;      // (the Viterbi pass sweeps band_in: TableView::ldm / stm)
  q.okbits_end = a.okbits_end ? a.okbits_end + p.bits_base : nullptr;
  v.row = a.seq_out + (size_t)n * a.out_stride;
  v.zs = a.zs + (size_t)g * 4;
}

// XCD-aware block swizzle (speed only): blocks are dealt round-robin over the 8 XCDs, each with a private L2.  The
// remap gives every XCD a contiguous range of virtual block ids, so the workgroups of one sequence -- which re-read each
// other's rows (out B segments of H1 / H2, neighbouring cells of the unary phase) -- share an L2.  Bijective for any grid.
__device__ __forceinline__ void swizzled_block(unsigned& bx, unsigned& by) {
  const unsigned nbx = gridDim.x, total = gridDim.x * gridDim.y;
  const unsigned orig = blockIdx.y * nbx + blockIdx.x;
  const unsigned q = total / 8, r = total % 8, xcd = orig % 8;
  const unsigned vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
  bx = __builtin_amdgcn_readfirstlane(vb % nbx);   // (uniform: keeps the block coordinates in scalar registers)
  by = __builtin_amdgcn_readfirstlane(vb / nbx);
}

// copies the automaton layout record into LDS (all threads; caller synchronises before the first use)
__device__ __forceinline__ void stage_layout(const LinArgs& a, AutomatonLayout* dst, int nthreads) {
  const int32_t* src = reinterpret_cast<const int32_t*>(a.layp);
  int32_t* d32 = reinterpret_cast<int32_t*>(dst);
  for (int t = threadIdx.x; t < (int)(sizeof(AutomatonLayout) / sizeof(int32_t)); t += nthreads) d32[t] = src[t];
}

// ---- exp(lambda_k * term) of all structural terms and loop items of the batch, once per evaluation
__global__ __launch_bounds__(kThreads) void k4_weights(LinWeightArgs a) {
  const ParamBlock* pb = reinterpret_cast<const ParamBlock*>(a.params);
  const double l0 = pb->lambda[0], l1 = pb->lambda[1];
  const size_t stride = (size_t)gridDim.x * kThreads;
  const size_t c_lo = a.cell_first, c_hi = a.cell_count ? a.cell_first + a.cell_count : a.n_cells;
  for (size_t c = c_lo + (size_t)blockIdx.x * kThreads + threadIdx.x; c < c_hi; c += stride) {
    const double e0 = a.e_stack[c], e1 = a.e_ext[c], e2 = a.e_ml[c], e3 = a.e_close[c], e4 = a.e_hp[c];
    a.xwc[0 * a.n_cells + c] = lin_weight(l0, e0); a.xwc[5 * a.n_cells + c] = lin_weight(l1, e0);
    a.xwc[1 * a.n_cells + c] = lin_weight(l0, e1); a.xwc[6 * a.n_cells + c] = lin_weight(l1, e1);
    a.xwc[2 * a.n_cells + c] = lin_weight(l0, e2); a.xwc[7 * a.n_cells + c] = lin_weight(l1, e2);
    a.xwc[3 * a.n_cells + c] = lin_weight(l0, e3); a.xwc[8 * a.n_cells + c] = lin_weight(l1, e3);
    a.xwc[4 * a.n_cells + c] = lin_weight(l0, e4); a.xwc[9 * a.n_cells + c] = lin_weight(l1, e4);
  }
  if (a.xwi)   // (the band kernels exponentiate the item terms themselves; an array only for callers that ask for one)
  for (size_t n = (size_t)blockIdx.x * kThreads + threadIdx.x; n < a.n_items; n += stride) {
    const double t = a.items[n].tsc;
    a.xwi[n] = lin_weight(l0, t);
    a.xwi[a.n_items + n] = lin_weight(l1, t);
    if (a.items_inner) {   // the same weights in the three secondary orders
      const double t1 = a.items_inner[n].tsc, t2 = a.items_left[n].tsc, t3 = a.items_right[n].tsc;
      a.xwi[2 * a.n_items + n] = lin_weight(l0, t1); a.xwi[3 * a.n_items + n] = lin_weight(l1, t1);
      a.xwi[4 * a.n_items + n] = lin_weight(l0, t2); a.xwi[5 * a.n_items + n] = lin_weight(l1, t2);
      a.xwi[6 * a.n_items + n] = lin_weight(l0, t3); a.xwi[7 * a.n_items + n] = lin_weight(l1, t3);
    }
  }
}

// LDS layout shared by k4_in / k4_out (host computes the size, kernels the pointers): nd doubles of accumulators and
// staging buffers come first, then the per-workgroup context: linear parameter block, exp position weights of the
// window of positions the workgroup touches, small int arrays, the automaton blob, and the dmin / base / unpaired
// windows.  With the context in LDS the unary phase has a single level of global loads (the tables themselves).
struct BlockLds {
  int lin, ews, ints, dm, cnts, pre, base, blob, bits, bits2, dmin16, seq8, unp8, crec, crfl, total;   // byte offsets
};
// LDS doubles of the item-record area of k4_in / k5_cyk (one role) and of k4_out (three roles)
constexpr int kRecIn = ELEMDP_RECIN, kRecOut = ELEMDP_RECOUT;
// Walks the set bits bit0 + n, n in [lo, hi], of a pair mask (ascending): next() returns n, or -1 at the end.
struct BitIter {
  const uint32_t* m;
  int bit0, e, w;
  uint32_t word;
  __device__ __forceinline__ void init(const uint32_t* mask, int b0, int lo, int hi) {
    m = mask; bit0 = b0; e = b0 + hi;
    const int b = b0 + (lo > 0 ? lo : 0);
    w = b >> 5;
    word = (hi >= lo) ? (mask[w] & (~0u << (b & 31))) : 0u;
    if (hi < lo) e = -1;
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
// crd: doubles per cell record of the table-driven unary phase (lin_fast.h), 0 = none
__host__ __device__ inline BlockLds block_lds(int nd, int cpb, int n_lin, int win, int n_stage, int nv = 0, int crd = 0) {
  BlockLds b;
  if (nv < cpb) nv = cpb;   // CSR ranges of the item sums: one per cell, or one per (role, cell) in k4_out
  int o = nd * 8;
  b.lin = o; o += n_lin * 8;
  b.ews = o; o += win * 8;
  b.crec = o; o += cpb * crd * 8;
  b.crfl = o; o += crd ? ((cpb + 1) / 2) * 8 : 0;
  b.ints = o;
  b.dm = o; o += cpb * 4;
  b.cnts = o; o += nv * 4;
  b.pre = o; o += (nv + 1) * 4;
  b.base = o; o += nv * 4;
  b.blob = o; o += n_stage * 4;
  // pair-mask words of the rows i0-1 .. i0+cpb+W (k4_out also reads the stems that start at the cells' ends); win = cpb + W + 3
  b.bits = o; o += ((win * (win - cpb - 2) + 31) / 32 + 3) * 4;
  b.bits2 = o; o += (((cpb + 1) * (win - cpb - 2) + 31) / 32 + 3) * 4;   // end-indexed mask, rows i0+d .. i0+d+cpb (k4_in)
  b.dmin16 = o; o += ((win + 1) / 2) * 4;
  b.seq8 = o; o += ((win + 3) / 4) * 4;
  b.unp8 = o; o += ((win + 3) / 4) * 4;
  b.total = o + 8;
  return b;
}
struct BlockCtx { int* dm; int* cnts; int* pre; int* base; };
// accumulator / staging doubles of k4_out: four heavy sums per (cell, state), the statistics of the two worlds (nw copies: one per
// wave in the deterministic mode), the position
// posteriors of the scan over the window of positions the workgroup touches (start + inner, or end), the item records
__host__ __device__ inline int out_doubles(int CS, int nt, int win, int nw = 1, bool scan = true) { return 4 * CS + nw * (2 * nt + 4) + (scan ? 2 * win : 0) + kRecOut; }
// ints of the automaton blob a band kernel stages: everything (n_stage = n_ints) means the small part plus the run of
// tuple lists of its direction (PART 0: inside, 1: outside); otherwise only the small part
__host__ __device__ inline int staged_ints(const AutomatonLayout& L, int n_stage, int part) {
  if (n_stage < L.n_ints) return n_stage;
  return L.n_small + (part == 0 ? L.big_in_end - L.n_small : L.n_ints - L.big_in_end);
}

// stages the context and redirects the views to it; positions [p0, p0+len) = [i0-1, i0+nc+d] clipped to [0, L].
// FAST: only the fast blob of the direction is staged (AutomatonLayout::fb_*); the generic lists stay in global memory.
template <bool BIG, int PART, bool FAST = false>
__device__ __forceinline__ BlockCtx stage_context(const LinArgs& a, LViews& v, unsigned char* raw, const BlockLds& B, int i0, int nc,
                                                  int d, int cpb) {
  const int tid = threadIdx.x;
  const int L = v.q.L;
  // (the outside kernel also needs dmin of the rows i - b, b <= W - d, whose pair entries feed HA: lheavy_o2)
  const int back = (PART == 1 && v.q.W - d > 1) ? v.q.W - d : 1;
  const int p0 = (i0 > back) ? i0 - back : 0;
  const int p1 = (i0 + nc + d < L) ? i0 + nc + d : L;   // inclusive
  const int len = p1 - p0 + 1;
  double* llin = reinterpret_cast<double*>(raw + B.lin);
  double* lews = reinterpret_cast<double*>(raw + B.ews);
  int* blob = reinterpret_cast<int*>(raw + B.blob);
  int16_t* ldmin = reinterpret_cast<int16_t*>(raw + B.dmin16);
  uint8_t* lseq = raw + B.seq8;
  uint8_t* lunp = raw + B.unp8;
  const int n_lin = a.n_lin;
  const int big_lo = FAST ? (PART == 0 ? a.lay.fb_in : a.lay.fb_out) : (PART == 0) ? a.lay.n_small : a.lay.big_in_end;
  const int big_hi = FAST ? big_lo + (PART == 0 ? a.lay.fb_in_n : a.lay.fb_out_n) : (PART == 0) ? a.lay.big_in_end : a.lay.n_ints;
  const int big_at = FAST ? 0 : a.lay.n_small;      // where the run lands in the LDS blob
  uint32_t* lbits = reinterpret_cast<uint32_t*>(raw + B.bits);
  uint32_t* lbits2 = reinterpret_cast<uint32_t*>(raw + B.bits2);
  // pair mask: bits of the cells (i0-1, .) .. (i0+nc-1, .)  (bit index = i * (W+1) + d); the outside kernel also walks the
  // stems that START at the ends j = i + d of its cells (rule 2, factorised): rows up to i0+nc-1+d.  The inside kernel
  // walks the stems that END there: rows i0+d .. i0+nc-1+d of the end-indexed mask (second window).
  const int W1 = v.q.W + 1;
  // (the inside kernel tests the inner pair (i+1, d-2) of its last cell: one row more)
  const int last_row = (PART == 1) ? ((i0 + nc + d <= L + 1) ? i0 + nc + d : L + 1) : ((i0 + nc + 1 <= L + 1) ? i0 + nc + 1 : L + 1);
  const int bit0 = ((i0 > 0) ? i0 - 1 : 0) * W1, bit1 = last_row * W1;   // [bit0, bit1)
  const int w0 = bit0 >> 5, w1 = ((bit1 + 31) >> 5) + 1;
  const int wend = (int)((((long long)(L + 1) * W1) + 31) >> 5);
  const int e0 = ((i0 + d) * W1) >> 5, e1 = (PART == 0) ? ((((i0 + d + nc) * W1 + 31) >> 5) + 1) : e0;
  // ALL loads of the context are issued before the first LDS store (clamped addresses, fixed unrolling): one round trip
  // for the whole context instead of one per array (a plain copy loop waits for its loads before it stores)
  constexpr int kU = 4;
  const int n_sm = FAST ? 0 : BIG ? a.lay.n_small : a.n_stage, n_bg = BIG ? big_hi - big_lo : 0;
  int r_sm[kU], r_bg[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int t = tid + u * kBT;
    r_sm[u] = a.ints[t < n_sm ? t : 0];
    r_bg[u] = a.ints[t < n_bg ? big_lo + t : 0];
  }
  const double r_lin = a.lin[tid < n_lin ? tid : 0];
  const uint32_t r_bits = v.q.okbits[(tid < w1 - w0 && w0 + tid < wend) ? w0 + tid : 0];
  const uint32_t r_bits2 = (PART == 0) ? v.q.okbits_end[(tid < e1 - e0 && e0 + tid < wend) ? e0 + tid : 0] : 0u;
  const int pw = p0 + (tid < len ? tid : 0);
  const double r_ews = v.q.ews[pw];
  const int16_t r_dmin = v.q.dmin[pw];
  const uint8_t r_unp = v.q.unp[pw];
  const uint8_t r_seq = v.q.seq[pw < L ? pw : (L > 0 ? L - 1 : 0)];
  const int16_t r_dm = v.q.dmin[i0 + (tid < nc ? tid : 0)];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int t = tid + u * kBT;
    if (t < n_sm) blob[t] = r_sm[u];
    if (t < n_bg) blob[big_at + t] = r_bg[u];
  }
  if (tid < n_lin) llin[tid] = r_lin;
  if (tid < w1 - w0) lbits[tid] = (w0 + tid < wend) ? r_bits : 0u;
  if (PART == 0 && tid < e1 - e0) lbits2[tid] = (e0 + tid < wend) ? r_bits2 : 0u;
  if (tid < len) {
    lews[tid] = r_ews;
    ldmin[tid] = r_dmin;
    lunp[tid] = r_unp;
    lseq[tid] = (pw < L) ? r_seq : (uint8_t)0;
  }
  // (larger automata / windows than the unrolled part covers)
  for (int t = tid + kU * kBT; t < n_sm; t += kBT) blob[t] = a.ints[t];
  for (int t = tid + kU * kBT; t < n_bg; t += kBT) blob[big_at + t] = a.ints[big_lo + t];
  for (int t = tid + kBT; t < n_lin; t += kBT) llin[t] = a.lin[t];
  for (int t = tid + kBT; t < w1 - w0; t += kBT) lbits[t] = (w0 + t < wend) ? v.q.okbits[w0 + t] : 0u;
  if (PART == 0) for (int t = tid + kBT; t < e1 - e0; t += kBT) lbits2[t] = (e0 + t < wend) ? v.q.okbits_end[e0 + t] : 0u;
  for (int t = tid + kBT; t < len; t += kBT) {
    const int p = p0 + t;
    lews[t] = v.q.ews[p];
    ldmin[t] = v.q.dmin[p];
    lunp[t] = v.q.unp[p];
    lseq[t] = (p < L) ? v.q.seq[p] : (uint8_t)0;
  }
  BlockCtx c;
  c.dm = reinterpret_cast<int*>(raw + B.dm);
  c.cnts = reinterpret_cast<int*>(raw + B.cnts);
  c.pre = reinterpret_cast<int*>(raw + B.pre);
  c.base = reinterpret_cast<int*>(raw + B.base);
  if (tid < cpb) c.dm[tid] = (tid < nc) ? (int)r_dm : 0;
  if (!FAST) {
    v.m.ints = blob;
    v.in.cm = v.out.cm = blob + a.lay.tab_cmap;
  }
  if (BIG) v.m.big = blob + big_at - big_lo;   // (indices of the staged run keep their global values)
  v.m.lin = llin;
  v.q.ews = lews - p0;
  v.q.dmin = ldmin - p0;
  v.q.unp = lunp - p0;
  v.q.seq = lseq - p0;
  v.q.okbits = lbits - w0;
  if (PART == 0) v.q.okbits_end = lbits2 - e0;
  return c;
}

// Heavy sums are evaluated from LDS: for a chunk of split points the operand rows of ALL cells of the workgroup are
// staged with one coalesced, fully unrolled batch of loads (consecutive cells of a diagonal are contiguous in the
// [e][d][i][s] layout: one segment of nc*S doubles per (plane, split point)), then one lane per (cell, state tuple)
// multiplies out of LDS.  Global loads are thereby independent of the tuple structure (every element is fetched
// once per workgroup, 2*kChunk loads in flight per lane) and the dependent chain per diagonal stays short.
constexpr int kChunkIn = ELEMDP_KCI;    // split points per staging round of k5_cyk (2 segments each)

// Which element of a staged operand row a lane copies.  Only the first NU = n_front states of a row can be non-zero in the
// planes of the bifurcation rule (B, 1, 2) of a complete parse (Automaton::flatten puts them first), so a staged row is
// nc * NU doubles instead of nc * S, and the lanes that are left over take further split points of the same round: lane =
// (sub, cell c, state k), `nsub` split points per load instruction, kChunk * nsub per staging round.  LDS image of one
// split point: [c * NU + k] (CU = cpb * NU doubles).
struct StageMap {
  int NU, CU, nsub, sub, r, goff, cell;
  bool act;
};
__device__ __forceinline__ StageMap stage_map(int n_front, int S, int cpb, int nc, int tid) {
  StageMap m;
  m.NU = n_front;
  m.CU = cpb * n_front;
  m.nsub = kBT / m.CU;            // >= 1: cpb * S <= kBT
  m.sub = tid / m.CU;
  m.r = tid - m.sub * m.CU;
  m.cell = m.r / n_front;
  const int k = m.r - m.cell * n_front;
  m.act = m.sub < m.nsub && m.cell < nc;
  m.goff = m.act ? m.cell * S + k : 0;
  return m;
}

// Item sums without serial chains: the interior-loop items of all cells of the workgroup (CSR ranges given by
// `range(c, &first, &last)`) form one flat list; work item = (item, state tuple), tuple fastest, so the lanes of a wave
// read the same three table rows.  Every lane takes kItemBatch work items at a time and runs them stage by stage
// (index -> item record -> table operands -> product), so that the dependent loads of one work item overlap with those
// of the others.  Loads of a stage are unconditional (work items past the end are clamped to the last one and dropped
// in the final stage).  `cnts`, `pre` (cpb+1) and `base` are LDS ints.
constexpr int kItemBatch = ELEMDP_KIB;
struct ItemSlot {
  int c, n, t, idx;   // cell of the workgroup, position in the CSR order, state tuple, item index
  LoopItem it;
  double x0, x1, x2, xw, aux;
  bool ok;
};
template <class RangeFn, class F1, class F2, class F3, class F4>
__device__ __forceinline__ void for_block_items(int nv, int nq, int tid, int* cnts, int* pre, int* base, RangeFn range,
                                                F1 load_index, F2 load_item, F3 load_operands, F4 finish) {
  // nv "virtual cells" (cell x role), each with one CSR range
  for (int vc = tid; vc < nv; vc += kBT) {
    int c0 = 0, c1 = 0;
    range(vc, c0, c1);
    base[vc] = c0;
    cnts[vc] = (c1 > c0) ? c1 - c0 : 0;
  }
  __syncthreads();
  for (int vc = tid; vc <= nv; vc += kBT) {
    int p = 0;
    for (int c = 0; c < vc; ++c) p += cnts[c];
    pre[vc] = p;
  }
  __syncthreads();
  const int total = pre[nv] * nq;
  for (int w0 = tid; w0 < total; w0 += kItemBatch * kBT) {
    ItemSlot sl[kItemBatch];
#pragma unroll
    for (int u = 0; u < kItemBatch; ++u) {
      const int w = w0 + u * kBT;
      sl[u].ok = w < total;
      const int wc = sl[u].ok ? w : total - 1;
      const int il = wc / nq;
      sl[u].t = wc - il * nq;
      int lo = 0, hi = nv - 1;   // the virtual cell owning item il: largest vc with pre[vc] <= il
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pre[mid] <= il) lo = mid; else hi = mid - 1;
      }
      sl[u].c = lo;
      sl[u].n = base[lo] + (il - pre[lo]);
    }
#pragma unroll
    for (int u = 0; u < kItemBatch; ++u) load_index(sl[u]);
#pragma unroll
    for (int u = 0; u < kItemBatch; ++u) load_item(sl[u]);
#pragma unroll
    for (int u = 0; u < kItemBatch; ++u) load_operands(sl[u]);
#pragma unroll
    for (int u = 0; u < kItemBatch; ++u) if (sl[u].ok) finish(sl[u]);
  }
  __syncthreads();
}

// (record, tuple) of the work item w = x * nq + t without a division per work item: the kernels are bound by instruction
// issue (SQ counters: the waves of a SIMD are active ~95 % of the time between them), and an integer division is ~25
// instructions.  `step` advances by kBT work items; q256 = kBT / nq, r256 = kBT % nq.
struct WorkIdx {
  int x, t;
  __device__ __forceinline__ void step(int q256, int r256, int nq) {
    x += q256;
    t += r256;
    if (t >= nq) { t -= nq; ++x; }
  }
};

// Item sums (rule 6c): a lane holds ONE item record and walks the tuples of the rule through their column records
// (AutomatonLayout::qc_*, staged with the tuple lists); the four waves share the tuples (wave w takes t = w, w + 4, ..), kTU of
// them per pass with all their table operands in flight together.  (Two lanes per record -- every tuple of the usual lists in
// one pass -- computes the record's row addresses twice and measured 2 % slower.)  Everything that depends
// on the record (row addresses, weights) is computed once per lane, a tuple then costs its four loads and a handful of
// instructions.
// Register-pressure analysis only (tools/kmeta.py ... -DELEMDP_KO=<bits>): phases compiled out of k4_in / k4_out.
// bit 0 pair phases, 1 item sums, 2 unary phase, 3 pair entries behind the unary phase (k4_out).  Results are invalid.
#ifndef ELEMDP_KO
#define ELEMDP_KO 0
#endif
#ifndef ELEMDP_KSTEMS
#define ELEMDP_KSTEMS 6
#endif
constexpr int kStems = ELEMDP_KSTEMS;   // stems of the factorised rule 2 whose operands a lane has in flight together (train kernels)
#ifndef ELEMDP_AHEAD_IN
#define ELEMDP_AHEAD_IN 0
#endif
#ifndef ELEMDP_AHEAD_OUT
#define ELEMDP_AHEAD_OUT 0
#endif
#ifndef ELEMDP_KTU_IN
#define ELEMDP_KTU_IN 3
#endif
#ifndef ELEMDP_KTU_OUT
#define ELEMDP_KTU_OUT 3
#endif
constexpr int kTUin = ELEMDP_KTU_IN, kTUout = ELEMDP_KTU_OUT;

// Item records of the workgroup's cells in LDS (k4_in, k5_cyk: the by_outer order; k4_out stages its three roles the same
// way inline): CSR range per cell -> prefix in LDS, then all lanes fetch the records of [p0, p0 + cap) with one round of
// loads.  meta = cell << 16 | position of the item in its cell; xw = exp(lambda_k tsc) for k = 0, 1 (0 when the item is
// not in the inside set).
struct OuterRecs { LoopItem* it; double* xw; int* meta; int cap; };
__device__ __forceinline__ OuterRecs outer_recs(double* area, int n_doubles) {
  OuterRecs r;
  r.cap = (n_doubles * 8) / 40;
  r.it = reinterpret_cast<LoopItem*>(area);
  r.xw = reinterpret_cast<double*>(r.it + r.cap);
  r.meta = reinterpret_cast<int*>(r.xw + 2 * r.cap);
  return r;
}
// ranges + prefix; returns the number of records of the workgroup
// (two halves: the CSR loads are issued before the pair phase of the kernel, whose own loads they travel with; the prefix
// follows behind the barrier that ends the pair phase)
__device__ __forceinline__ void outer_ranges_load(const LViews& v, int i0, int nc, int d, bool on, int tid, int* cnts, int* base) {
  for (int c = tid; c < nc; c += kBT) {
    const int i = i0 + c;
    int n0 = 0, n1 = 0;
    if (on && v.q.e_ok(i, d)) { const int cell = v.q.cell(i, d); n0 = v.q.by_outer_off[cell]; n1 = v.q.by_outer_off[cell + 1]; }
    base[c] = n0;
    cnts[c] = (n1 > n0) ? n1 - n0 : 0;
  }
}
// exclusive prefix of the LDS ints cnts[0 .. n) into pre[0 .. n], pre[n] = total (no barrier).  Up to 64 entries -- the CSR ranges
// of a workgroup's cells -- by one wave scan: a lane that adds up its predecessors itself walks a chain of n dependent LDS reads
// (~1 us for the 36 ranges of k4_out, in every workgroup).
__device__ __forceinline__ void lds_prefix(int n, int tid, const int* cnts, int* pre) {
  if (n <= 64) {
    if (tid < 64) {
      const int x = tid < n ? cnts[tid] : 0;
      int incl = x;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o, 64);
        if (tid >= o) incl += y;
      }
      if (tid < n) pre[tid] = incl - x;
      if (tid == 63) pre[n] = incl;
    }
    return;
  }
  for (int c = tid; c <= n; c += kBT) {
    int p = 0;
    for (int k = 0; k < c; ++k) p += cnts[k];
    pre[c] = p;
  }
}
__device__ __forceinline__ int outer_ranges_prefix(int nc, int tid, const int* cnts, int* pre) {
  lds_prefix(nc, tid, cnts, pre);
  __syncthreads();
  return pre[nc];
}
__device__ __forceinline__ int outer_ranges(const LViews& v, int i0, int nc, int d, bool on, int tid, int* cnts, int* pre, int* base) {
  outer_ranges_load(v, i0, nc, d, on, tid, cnts, base);
  __syncthreads();
  return outer_ranges_prefix(nc, tid, cnts, pre);
}
// record p of the workgroup (prefix `pre` over its nc cells): its cell and its index in the item arrays
__device__ __forceinline__ void outer_locate(int p, int nc, const int* pre, const int* base, int& lo, int& n) {
  lo = 0;
  int hi = nc - 1;   // the cell owning record p: largest c with pre[c] <= p
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pre[mid] <= p) lo = mid; else hi = mid - 1;
  }
  n = base[lo] + (p - pre[lo]);
}
// A record fetched ahead of the heavy sums (its loads travel with theirs): record `tid` of the workgroup
struct RecAhead { LoopItem it; int meta; bool have; };
__device__ __forceinline__ RecAhead outer_fetch_ahead(const LViews& v, int n_rec, int nc, int tid, const int* pre, const int* base) {
  RecAhead r;
  r.have = tid < n_rec;
  r.meta = 0;
  r.it = LoopItem{0., 0, 0, 0, 0};
  if (r.have) {
    int lo, n;
    outer_locate(tid, nc, pre, base, lo, n);
    const bool in = v.q.item_in[n] != 0;
    r.it = v.q.items[n];
    r.meta = (in ? 0 : (int)0x80000000) | (lo << 16) | (tid - pre[lo]);
  }
  return r;
}
template <bool WEIGHTS, bool AHEAD = false>
__device__ __forceinline__ void outer_stage(const LViews& v, const OuterRecs& r, int p0, int np, int nc, int tid, const int* pre,
                                            const int* base, const RecAhead ahead = RecAhead{LoopItem{0., 0, 0, 0, 0}, 0, false}) {
  for (int x = tid; x < np; x += kBT) {
    const int p = p0 + x;
    LoopItem itv;
    int meta;
    if (AHEAD && p == tid && ahead.have) { itv = ahead.it; meta = ahead.meta; }
    else {
      int lo, n;
      outer_locate(p, nc, pre, base, lo, n);
      const bool in = v.q.item_in[n] != 0;
      itv = v.q.items[n];
      meta = (in ? 0 : (int)0x80000000) | (lo << 16) | (p - pre[lo]);
    }
    r.it[x] = itv;
    if (WEIGHTS) {   // exp(lambda_k * tsc) on the spot: two exps per staged record are cheaper than an array of them in HBM
      const double tsc = itv.tsc;
      r.xw[x] = meta >= 0 ? lin_weight(v.m.lambda[0], tsc) : 0.;
      r.xw[r.cap + x] = meta >= 0 ? lin_weight(v.m.lambda[1], tsc) : 0.;
    }
    r.meta[x] = meta;
  }
  __syncthreads();
}

// FAST: table-driven phases (lin_fast.h; train schedule, the fast blob staged); FP: longest pair list of a state (2 or 3)
// MB: the workgroup owns several blocks of cells (LinArgs::nblk > 1) -- a loop around the phases that costs registers (values that
// are the same in every block stay live around it), so the one-block form is a kernel of its own: small groups and the scan
template <bool BIG, bool CON, bool FAST = false, int FP = kFastP, bool W8 = false, bool MB = false>
__global__ __launch_bounds__(kBT, W8 ? ELEMDP_LB_IN_FAST : ELEMDP_LB_IN) void k4_in(LinArgs a) {
  extern __shared__ double lds[];
  // (the automaton layout is read from the kernel arguments: constant offsets, scalar registers)
  PhaseClock pc;
  pc.start(a.prof);

  unsigned bx, by;
  swizzled_block(bx, by);
  LViews v(a.lay);
  make_lviews(a, by, v);
  const AutomatonLayout& A = a.lay;
  const int S = a.lay.S, NA = a.lay.n_active, d = a.d, cpb = a.cpb, tid = threadIdx.x;
  // The workgroup owns nblk blocks of cpb consecutive cells of the diagonal (cpbT cells from i0T on): everything that does not
  // depend on the block -- plan record, automaton blob, parameters and weight tables, the window of positions, the cell records
  // and CSR ranges of all its cells -- is fetched and staged ONCE; the phases then run block by block on the per-block heavy
  // sums.  (One block per workgroup paid the launch, two dependent round trips and the staging for 12 cells of work.)
  const int nblk = (MB && a.nblk > 1) ? a.nblk : 1, cpbT = cpb * nblk;
  if (d > v.q.W) return;
  const int ncell = v.q.L - d + 1, i0T = bx * cpbT;
  if (i0T >= ncell) return;
  const int ncT = (cpbT < ncell - i0T) ? cpbT : ncell - i0T;
  if (CON) {
    // The start constraint touches the emissions of position Ys only: a cell whose span does not cover Ys -- and everything below
    // it -- has the value of the unconstrained pass, which is still in the table (same slot, same layout; the scan runs
    // k4_in<false> first).  Only the cells i <= Ys < i + d are swept again: ~d of the L - d + 1 cells of the diagonal.
    const int ys = a.ys[v.n];
    if (ys < i0T || ys - d + 1 > i0T + ncT - 1) return;
  }
  const int HD = FAST ? A.n_lane : S;   // stride of the heavy sums per cell: the live states (table-driven: their index among them), or all
  const int CS = cpb * HD;
  constexpr int NW = 1;                      // (one copy of the heavy sums in either mode: see the deterministic mode above)
  double* hb = lds;
  double* he = hb + CS;
  double* hbA = hb;
  double* heA = he;
  double* st1 = lds + NW * 2 * CS;           // item records (kRecIn doubles)
  const BlockLds BL = block_lds(NW * 2 * CS + kRecIn, cpbT, a.n_lin, cpbT + a.wmax + 3, FAST ? a.lay.fb_in_n : staged_ints(a.lay, a.n_stage, 0), 0, FAST ? kCellInD : 0);
  // cell records of the table-driven unary phase: the exponentiated structural terms of the cells are fetched with the context
  // (lane = (cell, value); the addresses depend on the plan record only), the flags follow once the context is in LDS
  constexpr int kCRin = (ELEMDP_CPB_MAX * 8 + kBT - 1) / kBT;
  double crx[kCRin];
  if (FAST) {
#pragma unroll
    for (int r = 0; r < kCRin; ++r) {
      crx[r] = 0.;
      if (r * kBT < cpbT * 8) {   // (uniform: most automata need the first round only)
        const int t = tid + r * kBT, c = (t >> 3) < ncT ? (t >> 3) : 0;
        crx[r] = cell_in_fetch(v.q, d, i0T + c, t & 7);
      }
    }
  }
  if (a.dbg & 1024) { if (ncT == 12345) lds[tid] = crx[0]; return; }   // (timing experiments: the workgroup up to its plan record ...
  const BlockCtx cx = stage_context<BIG, 0, FAST>(a, v, reinterpret_cast<unsigned char*>(lds), BL, i0T, ncT, d, cpbT);
  int* dmT = cx.dm; int* cntsT = cx.cnts; int* pre = cx.pre; int* baseT = cx.base;
  const int32_t* G = v.m.big;
  double* crecT = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(lds) + BL.crec);
  int* crflT = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(lds) + BL.crfl);
  for (int t = tid; t < NW * 2 * CS; t += kBT) lds[t] = 0.;
  __syncthreads();
  if (a.dbg & 2048) return;                                             //  ... and up to its staged context)
  pc.mark<0>();
  if (FAST) {   // (read by the unary phase, behind the barriers of the heavy sums)
#pragma unroll
    for (int r = 0; r < kCRin; ++r) {
      const int t = tid + r * kBT, c = t >> 3, k = t & 7;
      if (c < ncT) {
        const bool on = k < 4 ? v.q.pair_ok(i0T + c, d) : v.q.e_ok(i0T + c, d);
        crecT[c * kCellInD + 2 + k] = on ? crx[r] : 0.;
      }
    }
    for (int c = tid; c < ncT; c += kBT) {
      const int i = i0T + c, j = i + d;
      int fl = cell_in_flags(v.m, v.q, d, i);
      if (CON) { const int ys = a.ys[v.n]; fl |= (i == ys ? CF_YL : 0) | (j - 1 == ys ? CF_YR : 0); }
      crflT[c] = fl;
      crecT[c * kCellInD] = v.q.ews[i];
      crecT[c * kCellInD + 1] = v.q.ews[j > 0 ? j - 1 : 0];
    }
  }
  const double* B = v.in.band;
  const int nq = ((a.dbg & 2) || (ELEMDP_KO & 2)) ? 0 : A.n_quad;
  // CSR ranges of the item sums of all cells of the workgroup (consumed behind the first pair phase, whose loads they travel with)
  outer_ranges_load(v, i0T, ncT, d, nq > 0, tid, cntsT, baseT);
  const RecAhead ahead{LoopItem{0., 0, 0, 0, 0}, 0, false};   // (records fetched ahead of the pair phase: two barriers more than the round trip saved, measured)
  const int tid_wg = tid;
  for (int blk = 0; MB ? blk * cpb < ncT : blk < 1; ++blk) {
  // (the lane id behind an empty asm: what a lane derives from it -- its cell, pair record, program, tuple records -- is the same in
  // every block, and the compiler would keep all of it in registers across the whole loop: 98 instead of 65 VGPRs)
  int tid = tid_wg;
  if (MB) asm volatile("" : "+v"(tid));
  const int i0 = i0T + blk * cpb, nc = (cpb < ncT - blk * cpb) ? cpb : ncT - blk * cpb;
  if (CON) {   // (uniform: blocks without a cell that covers Ys keep the values of the unconstrained pass)
    const int ys = a.ys[v.n];
    if (ys < i0 || ys - d + 1 > i0 + nc - 1) continue;
  }
  const int* dm = dmT + blk * cpb;
  int* cnts = cntsT + blk * cpb;
  int* base = baseT + blk * cpb;
  const double* crec = crecT + blk * cpb * kCellInD;
  const int* crfl = crflT + blk * cpb;
  // rule 2, factorised (lin_rules.h, lin_inside_apair): lane = (cell, pair p = (s1, t)).  A(i,j,p) = the tail step from
  // A(i,j-1,.) plus one term per stem (k, j) that ends at j and starts behind i; B(i,j,tgt(p)) += A(i,j,p).  The stems are
  // walked four at a time: their operand loads (1(i,k,s1), P(k,j,t), exp(lambda e_ml)) are in flight together.
  {
    const int nA = A.n_ap;
    const int32_t* I = v.m.ints;
    const Constraint con{CON ? a.ys[v.n] : -1, -1, 0};
    const int W1 = v.q.W + 1;
    // (deterministic mode: a cell's pairs take 2^det_sh lanes, so no cell straddles two waves; more than 64 pairs: wave 0 alone)
    const int pad = (a.det && a.det_sh >= 0) ? (1 << a.det_sh) : nA;
    const int nwork = ((a.dbg & 1) || (ELEMDP_KO & 1)) ? 0 : nc * pad;
    const bool one_wave = a.det && a.det_sh < 0;
    for (int w = one_wave ? ((tid < 64) ? tid : nwork) : tid; w < nwork; w += one_wave ? 64 : kBT) {
      const int c = (a.det && a.det_sh >= 0) ? (w >> a.det_sh) : div_rcp(w, a.rcp_nap), p = w - c * pad;
      if (p >= nA) continue;
      const int i = i0 + c, j = i + d;
      if (FAST) {   // the same sums from the pair record (AutomatonLayout::fpr_in): columns, chain entries with their weight ids
        const int32_t* PR = G + A.fpr_in + 8 * p;
        const int r0 = PR[0], r1 = PR[1];
        const int c1 = fcol(r0, 0), cP = fcol(r0, 1), tg = (r0 >> 16) & 0xff;
        const int dmi = dm[c];
        double av = 0.;
        if (dmi > 0 && dmi < d) {
          BitIter it;
          it.init(v.q.okbits_end, j * W1, 1, d - dmi);
          const int nch = (dmi < d - 1 && v.q.unp[j - 1]) ? (r1 >> 16) & 15 : 0;   // (the entries of (i, d-1) exist iff dmin[i] < d - 1)
          int ce[kFastR];
          double pv[kFastR];
#pragma unroll
          for (int u = 0; u < kFastR; ++u) {
            ce[u] = PR[2 + u];
            pv[u] = v.in.lda(d - 1, i, u < nch ? ce[u] & 0xff : p, u < nch);
          }
          bool first = true;
          for (;;) {
            int sp[kStems];
            sp[0] = it.next();
            if (sp[0] < 0 && !first) break;
#pragma unroll
            for (int u = 1; u < kStems; ++u) sp[u] = sp[u - 1] < 0 ? -1 : it.next();
            // (b = X(k,j,t) = P * exp(lambda e_ml), stored by the unary phase of the stem's diagonal under P's column in the B plane)
            double sa[kStems], sb[kStems];
#pragma unroll
            for (int u = 0; u < kStems; ++u) {
              const bool g = sp[u] >= 0;
              const int q = g ? sp[u] : 1;
              sa[u] = v.in.ldc(ST_1, d - q, i, c1, g);
              sb[u] = v.in.ldc(ST_B, q, j - q, cP, g);
            }
            if (first) {   // tail step (its operands travelled with the first stems')
              first = false;
              const double wt = (r0 & (2 << 24)) ? v.q.ews[j - 1] : 1.;
              const int bj = v.q.seq[j - 1];
#pragma unroll
              for (int u = 0; u < kFastR; ++u)
                if (u < nch) {
                  const int id = (ce[u] >> 8) & 0x7fff;
                  if (CON && j - 1 == con.ys && !(G[A.fs_in + id] & SF_SR)) continue;   // allow_right
                  av = fma(pv[u], v.m.lin[A.lin_wr + 5 * id + bj] * wt, av);
                }
            }
#pragma unroll
            for (int u = 0; u < kStems; ++u) av = fma(sa[u], sb[u], av);      // (0 * 0 for the slots without a stem)
            if (sp[kStems - 1] < 0) break;
          }
          v.in.a(d, i, p) = av;
        }
        if (tg != 0xff && av != 0.) atomicAdd(&hbA[c * HD + tg], av);
        continue;
      }
      const int s1 = I[A.ap_s1 + p], t = I[A.ap_t + p], tgt = I[A.ap_tgt + p];
      const int c1 = v.in.col(ST_1, s1), cP = v.in.col(ST_P, t);
      const int dmi = dm[c];
      double av = 0.;
      if (dmi > 0 && dmi < d) {          // 1(i,k,.) != 0 needs k - i >= dmin[i], and k < j: otherwise no entry (i, d, .) exists
        const double* xml = v.q.xwc + (size_t)(lamk(v.m, t) * 5 + XT_ML) * v.q.xwc_stride;
        BitIter it;
        it.init(v.q.okbits_end, j * W1, 1, d - dmi);      // spans of the stems: k = j - sp >= i + dmin[i]
        // tail step: the (at most kUnary, else looped) predecessors' values are in flight with the first stems
        const int e0 = I[A.ap_chain_off + p], ne = v.q.unp[j - 1] ? I[A.ap_chain_off + p + 1] - e0 : 0;
        double pv[kUnary];
        const bool tail = dmi < d - 1;   // (the entries of (i, d-1) exist)
#pragma unroll
        for (int u = 0; u < kUnary; ++u) pv[u] = v.in.lda(d - 1, i, u < ne ? I[A.ap_chain_ent + 2 * (e0 + u)] : p, tail);
        bool first = true;
        for (;;) {
          const int sp0 = it.next();
          if (sp0 < 0) {
            if (!first) break;
            // (no stem ends here: only the tail step)
          }
          if (first) {
            first = false;
#pragma unroll
            for (int u = 0; u < kUnary; ++u)
              if (u < ne) {
                const int pc2 = I[A.ap_chain_ent + 2 * (e0 + u)], tf = I[A.ap_chain_ent + 2 * (e0 + u) + 1];
                if (!CON || allow_right(v.m, con, v.q.L, j, t, I[A.ap_t + pc2])) av = fma(pv[u], lw_right(v.m, v.q, t, tf, j - 1), av);
              }
            for (int u = kUnary; u < ne; ++u) {
              const int pc2 = I[A.ap_chain_ent + 2 * (e0 + u)], tf = I[A.ap_chain_ent + 2 * (e0 + u) + 1];
              if (CON && !allow_right(v.m, con, v.q.L, j, t, I[A.ap_t + pc2])) continue;
              av = fma(v.in.lda(d - 1, i, pc2, tail), lw_right(v.m, v.q, t, tf, j - 1), av);
            }
            if (sp0 < 0) break;
          }
          const int sp1 = it.next(), sp2 = (sp1 < 0) ? -1 : it.next(), sp3 = (sp2 < 0) ? -1 : it.next();
          const int q1 = sp1 < 0 ? sp0 : sp1, q2 = sp2 < 0 ? sp0 : sp2, q3 = sp3 < 0 ? sp0 : sp3;
          // (1(i,k,.) and the stem P(k,j,.) are parsable by construction of the walk)
          const double a0 = v.in.ldc(ST_1, d - sp0, i, c1), b0 = v.in.ldc(ST_P, sp0, j - sp0, cP), w0 = xml[v.q.cell(j - sp0, sp0)];
          const double a1 = v.in.ldc(ST_1, d - q1, i, c1), b1 = v.in.ldc(ST_P, q1, j - q1, cP), w1 = xml[v.q.cell(j - q1, q1)];
          const double a2 = v.in.ldc(ST_1, d - q2, i, c1), b2 = v.in.ldc(ST_P, q2, j - q2, cP), w2 = xml[v.q.cell(j - q2, q2)];
          const double a3 = v.in.ldc(ST_1, d - q3, i, c1), b3 = v.in.ldc(ST_P, q3, j - q3, cP), w3 = xml[v.q.cell(j - q3, q3)];
          av = fma(a0, b0 * w0, av);
          if (sp1 >= 0) av = fma(a1, b1 * w1, av);
          if (sp2 >= 0) av = fma(a2, b2 * w2, av);
          if (sp3 >= 0) av = fma(a3, b3 * w3, av);
          if (sp3 < 0) break;
        }
        v.in.a(d, i, p) = av;
      }
      if (tgt >= 0 && av != 0.) atomicAdd(&hbA[c * HD + tgt], av);
    }
  }
  __syncthreads();
  pc.mark<1>();
  // rule 6c: E(i,j,tgt) += sum_items P(k,l,s1) * L(i,k,s2) * L(l,j,s3) * exp(lambda * tsc): work item = (item, tuple), the
  // item records staged in the (now free) operand staging area, one round of table loads per work item
  {
    const int n_rec = outer_ranges_prefix(nc, tid, cnts, pre);
    const OuterRecs R = outer_recs(st1, kRecIn);
    for (int p0 = 0; p0 < n_rec; p0 += R.cap) {
      const int np = (R.cap < n_rec - p0) ? R.cap : n_rec - p0;
      outer_stage<true, false>(v, R, p0, np, nc, tid, pre, base, ahead);
      const int wv = tid >> 6, lane = tid & 63;
      const int qc_in = FAST ? A.fqc_in : A.qc_in;
      constexpr int kTU = kTUin;
      for (int xb = 0; xb < np; xb += 64) {
        const int x = xb + lane;
        const int xc = x < np ? x : np - 1;
        const LoopItem it = R.it[xc];
        const int meta = R.meta[xc];
        const int c = (meta >> 16) & 0x7fff;
        const int i = i0 + c, j = i + d;
        const bool in_set = x < np && meta >= 0;   // (sign bit: not in the inside enumeration)
        // (the inner pair of an item is a kept pair, the loops L are stored everywhere)
        const uint32_t rP = v.in.cidx(ST_P, it.l - it.k, it.k, 0), rL1 = v.in.cidx(ST_L, it.k - i, i, 0), rL2 = v.in.cidx(ST_L, j - it.l, it.l, 0);
        const double xw0 = R.xw[xc], xw1 = R.xw[R.cap + xc];
        double* hrow = heA + c * HD;
        // the wave's tuples: every fourth one, or (deterministic mode) the ones whose targets are its own (qd_in: offsets, ids)
        const int qd = FAST ? A.fqd_in : A.qd_in;
        const int m0 = a.det ? __builtin_amdgcn_readfirstlane(G[qd + wv]) : 0;
        const int n_mine = a.det ? __builtin_amdgcn_readfirstlane(G[qd + wv + 1]) - m0 : (nq > wv ? (nq - wv + kWaves - 1) / kWaves : 0);
        for (int k0 = 0; k0 < n_mine; k0 += kTU) {
          int qa[kTU], qb[kTU];
          double x0[kTU], x1[kTU], x2[kTU];
#pragma unroll
          for (int u = 0; u < kTU; ++u) {
            const bool on = k0 + u < n_mine;
            const int kk = on ? k0 + u : 0;
            const int t = a.det ? __builtin_amdgcn_readfirstlane(G[qd + 5 + m0 + kk]) : wv + kWaves * kk;
            // (the tuple is the same for all lanes of the wave: its record in scalar registers, unpacked by the scalar unit)
            qa[u] = __builtin_amdgcn_readfirstlane(G[qc_in + 2 * t]);
            qb[u] = on ? __builtin_amdgcn_readfirstlane(G[qc_in + 2 * t + 1]) : (4 << 16);
            x0[u] = B[rP + (qa[u] & 0xff)]; x1[u] = B[rL1 + ((qa[u] >> 8) & 0xff)]; x2[u] = B[rL2 + ((qa[u] >> 16) & 0xff)];
          }
#pragma unroll
          for (int u = 0; u < kTU; ++u) {
            const double term = x0[u] * (x1[u] * x2[u]) * ((qb[u] & (1 << 16)) ? xw1 : xw0);
            if (!(qb[u] & (4 << 16)) && in_set && term != 0.) atomicAdd(&hrow[qb[u] & 0xffff], term);
          }
        }
      }
      __syncthreads();
    }
  }
  pc.mark<3>();
  const int NL = FAST ? A.n_lane : NA;   // lanes per cell of the unary phase: the states that have a column at all
  if (tid < nc * NL && !(a.dbg & 4) && !(ELEMDP_KO & 4)) {
    const int c = div_rcp(tid, a.rcp_lane);
    const int s = FAST ? G[A.f_live_in + tid - c * NL] : tid - c * NL;
    const int i = i0 + c;
    if (FAST) {
      fast_inside_unary<kFastR, FP, kFastL, CON>(A, G + A.fp_in + s * kFastW, v.m.lin, v.in, crec + c * kCellInD, crfl[c], d, i, hb + c * HD + (tid - c * NL),
                                                 he + c * HD + (tid - c * NL), NW, 2 * CS, G + A.fs_in);
      // (the lane owns these two sums: it clears them for the next block of the workgroup -- HD = NL here, and cells past nc take no adds)
      if (nblk > 1)
        for (int r = 0; r < NW; ++r) { hb[r * 2 * CS + tid] = 0.; he[r * 2 * CS + tid] = 0.; }
    } else {
      const Constraint con{CON ? a.ys[v.n] : -1, -1, 0};
      lin_inside_target_u<CON>(v.m, v.q, v.in, d, i, s, rep_sum(hb + c * HD + s, NW, 2 * CS), rep_sum(he + c * HD + s, NW, 2 * CS), con);
    }
  }
  pc.mark<4>();
  if (MB && (blk + 1) * cpb < ncT) {   // the next block's pair phase adds to the heavy sums this unary phase has read
    if (!FAST || (a.dbg & 4) || (ELEMDP_KO & 4)) {
      __syncthreads();
      for (int t = tid; t < NW * 2 * CS; t += kBT) lds[t] = 0.;
    }
    __syncthreads();
  }
  }   // blocks of the workgroup
  pc.finish();
}

// Exterior-chain kernels (one workgroup of 128 per sequence, L sequential steps): the per-sequence context of the whole
// sequence is staged in LDS when it fits (STAGE): pair mask, base codes, unpaired flags, position weights, automaton
// blob and linear parameters -- a step is then a handful of LDS reads plus the table rows of the few pairs at that
// position, instead of ~W dependent global loads for the pair tests alone.
struct ExtLds { int lin, ews, blob, bits, seq8, unp8, total; };
__host__ __device__ inline ExtLds ext_lds(int nd, int n_lin, int Lmax, int nword, int n_stage) {
  ExtLds b;
  int o = nd * 8;
  b.lin = o; o += n_lin * 8;
  b.ews = o; o += (Lmax + 1) * 8;
  b.blob = o; o += n_stage * 4;
  b.bits = o; o += (nword + 1) * 4;
  b.seq8 = o; o += ((Lmax + 4) / 4) * 4;
  b.unp8 = o; o += ((Lmax + 4) / 4) * 4;
  b.total = o + 8;
  return b;
}
// The exterior-chain kernels read the chain's own last W+1 rows at every step: those live in an LDS ring of a power-of-two
// number of rows (TableView::omask) beside the global table -- a step then waits for the band rows of its pairs only, not
// for the row the previous step has just stored.  0: the ring would not fit beside the staged context (wide bands, long
// sequences), the chain is read from global.  nd0: the kernel's own doubles in front (statistics).  Only for groups whose
// workgroups are all resident anyway (LinArgs::ext_ring): the ring halves the workgroups a CU holds, and a large group
// then takes two rounds of them (k4_in_ext +46 % at 3 333 sequences per launch, -7 % at 64).
__host__ __device__ inline int ext_ring_rows(int Wmax) { int r = 2; while (r < Wmax + 2) r <<= 1; return r; }
__host__ __device__ inline int ext_ring_doubles(int nd0, int Wmax, int S, int n_lin, int Lmax, int nword, int n_stage) {
  const long long n = (long long)ext_ring_rows(Wmax) * S;
  if (S > 128 || n * 8 > 48 * 1024) return 0;
  return ext_lds(nd0 + (int)n, n_lin, Lmax, nword, n_stage).total <= 64 * 1024 ? (int)n : 0;
}
template <bool STAGE>
__device__ __forceinline__ void stage_ext_context(const LinArgs& a, LViews& v, unsigned char* raw, int nd, int nthr = 128) {
  if (!STAGE) return;
  const int tid = threadIdx.x, L = v.q.L;
  const int nword = (int)((((long long)(L + 1) * (v.q.W + 1)) + 31) >> 5);
  const ExtLds B = ext_lds(nd, kLinEth + a.lay.n_theta, a.lmax, a.nword_max, a.n_stage);
  double* llin = reinterpret_cast<double*>(raw + B.lin);
  double* lews = reinterpret_cast<double*>(raw + B.ews);
  int* blob = reinterpret_cast<int*>(raw + B.blob);
  uint32_t* lbits = reinterpret_cast<uint32_t*>(raw + B.bits);
  uint8_t* lseq = raw + B.seq8;
  uint8_t* lunp = raw + B.unp8;
  for (int t = tid; t < a.n_stage; t += nthr) blob[t] = a.ints[t];
  for (int t = tid; t < kLinEth + a.lay.n_theta; t += nthr) llin[t] = a.lin[t];
  for (int t = tid; t < nword; t += nthr) lbits[t] = v.q.okbits[t];
  for (int t = tid; t <= L; t += nthr) {
    lews[t] = v.q.ews[t];
    lunp[t] = v.q.unp[t];
    lseq[t] = (t < L) ? v.q.seq[t] : (uint8_t)0;
  }
  v.m.ints = blob;
  v.in.cm = v.out.cm = blob + a.lay.tab_cmap;
  if (a.n_stage >= a.lay.n_ints) v.m.big = blob;
  v.m.lin = llin;
  v.q.ews = lews;
  v.q.unp = lunp;
  v.q.seq = lseq;
  v.q.okbits = lbits;
}

__device__ __forceinline__ bool out_of_range(double z) { return !(z > 0.) || !(z < HUGE_VAL); }

// ---- exterior chain of the inside pass, partition functions, objective (one workgroup of 128 per sequence)
constexpr int kExtBlock = 4;
// Launched with 128 threads, or with 512 = kExtBlock x 128 (LinArgs::ext_block = 4): the pair sums (rule 7) of four consecutive
// steps are then formed side by side -- a pair spans >= 5 positions, so they only read chain rows that are already there -- and
// only rule 8 and the sums of the parts stay sequential (LDS only): a quarter of the global round trips and of the long phases.
template <bool STAGE, bool CON>
__global__ __launch_bounds__(128 * kExtBlock) void k4_in_ext(LinArgs a) {
  extern __shared__ double l_ext[];
  __shared__ AutomatonLayout s_lay;
  const int NT = (int)blockDim.x, KB = NT / 128;   // (128 threads: the plain chain)
  stage_layout(a, &s_lay, NT);
  LViews v(s_lay);
  make_lviews(a, blockIdx.x, v);
  const int n_ring = (STAGE && a.ext_ring) ? ext_ring_doubles(0, a.wmax, a.lay.S, kLinEth + a.lay.n_theta, a.lmax, a.nword_max, a.n_stage) : 0;
  stage_ext_context<STAGE>(a, v, reinterpret_cast<unsigned char*>(l_ext), n_ring, NT);
  const int S = a.lay.n_active, tid = threadIdx.x, L = v.q.L;
  TableView Tr = v.in;      // the chain's rows through the LDS ring
  if (n_ring > 0) { Tr.ext = l_ext; Tr.omask = (uint32_t)ext_ring_rows(a.wmax) - 1u; }
  for (int s = tid; s < S; s += NT) {
    const double o0 = (s == a.lay.s00 || s == a.lay.shadow) ? 1. : 0.;   // (the shadow of (0,0) starts like it)
    v.in.o(0, s) = o0;
    if (n_ring > 0) Tr.o(0, s) = o0;
  }
  __syncthreads();
  // lanes = (part, state): the pairs ending at j are dealt to nparts lanes per state, the partial sums meet in LDS
  __shared__ double s_part[128 * kExtBlock];
  const int nparts = (S <= 128) ? 128 / S : 1;
  const int blk = tid >> 7, t128 = tid & 127;       // step of the block this lane sums (0 with 128 threads)
  const int part = t128 / S, ps = t128 - part * S;
  const Constraint con{CON ? a.ys[v.n] : -1, -1, 0};
  // The band values a step needs (the pairs that end at j) do not depend on the chain: the FIRST pair of a lane's share of step
  // j + 1 is fetched while step j is finished (ExtAhead), so a step with at most one pair per lane -- nearly all of them -- waits
  // for no global round trip.  Same terms in the same order as lin_inside_ext_part.
  constexpr int kAU = 4;
  struct ExtAhead { int i; int nu; double xe; double pv[kAU]; int e2[kAU]; };
  auto fetch_ahead = [&](int j, int s, int prt) {
    ExtAhead h;
    h.i = -1; h.nu = 0; h.xe = 0.;
#pragma unroll
    for (int k = 0; k < kAU; ++k) { h.pv[k] = 0.; h.e2[k] = 0; }
    if (j > L) return h;
    const AutomatonLayout& A = v.m.lay;
    const int32_t* G = v.m.big;
    const int i0 = (j - v.q.W > 0) ? j - v.q.W : 0;
    for (int i = j - 1 - prt; i >= i0; i -= nparts)
      if (v.q.pair_ok(i, j - i)) { h.i = i; break; }
    if (h.i < 0) return h;
    const int d = j - h.i, u0 = G[A.split_off + s];
    h.nu = G[A.split_off + s + 1] - u0;
    h.xe = xw_cell(v.q, lamk(v.m, s), XT_EXT, v.q.cell(h.i, d));
#pragma unroll
    for (int k = 0; k < kAU; ++k) {
      const int u = k < h.nu ? u0 + k : (h.nu > 0 ? u0 : 0);
      h.e2[k] = G[A.split_ent + 2 * u];
      h.pv[k] = Tr.ld(ST_P, d, h.i, G[A.split_ent + 2 * u + 1], h.nu > 0);
    }
    return h;
  };
  auto step_with = [&](const ExtAhead& h, int j, int s, int prt, bool rule8) {
    const AutomatonLayout& A = v.m.lay;
    const int32_t* I = v.m.ints;
    const int32_t* G = v.m.big;
    const int kl = lamk(v.m, s);
    double acc = 0.;
    if (h.i >= 0) {
      const int u0 = G[A.split_off + s];
      double b = 0.;
#pragma unroll
      for (int k = 0; k < kAU; ++k)
        if (k < h.nu) b = fma(Tr.o(h.i, h.e2[k]), h.pv[k], b);
      for (int u = u0 + kAU; u < u0 + h.nu; ++u)
        b = fma(Tr.o(h.i, G[A.split_ent + 2 * u]), Tr.ld(ST_P, j - h.i, h.i, G[A.split_ent + 2 * u + 1]), b);
      acc = fma(b, h.xe, acc);
      const int i0 = (j - v.q.W > 0) ? j - v.q.W : 0;
      for (int i = h.i - nparts; i >= i0; i -= nparts) {   // (further pairs of the lane's share: rare)
        const int d = j - i;
        if (!v.q.pair_ok(i, d)) continue;
        const double xe = xw_cell(v.q, kl, XT_EXT, v.q.cell(i, d));
        double b2 = 0.;
        for (int u = u0; u < u0 + h.nu; ++u)
          b2 = fma(Tr.o(i, G[A.split_ent + 2 * u]), Tr.ld(ST_P, d, i, G[A.split_ent + 2 * u + 1]), b2);
        acc = fma(b2, xe, acc);
      }
    }
    if (rule8 && prt == 0 && v.q.unp[j - 1])
      for (int t = I[A.right_off + s]; t < I[A.right_off + s + 1]; ++t) {
        if (CON && !allow_right(v.m, con, v.q.L, j, s, I[A.right_ent + 2 * t])) continue;
        acc = fma(Tr.o(j - 1, I[A.right_ent + 2 * t]), lw_right(v.m, v.q, s, I[A.right_ent + 2 * t + 1], j - 1), acc);
      }
    return acc;
  };
  if (KB > 1) {   // blocked chain (S <= 128, staged context: the launcher's condition)
    for (int j0 = 1; j0 <= L; j0 += KB) {
      const int ja = j0 + blk;
      if (part < nparts && ja <= L) s_part[tid] = step_with(fetch_ahead(ja, ps, part), ja, ps, part, false);
      __syncthreads();
      for (int b = 0; b < KB && j0 + b <= L; ++b) {
        const int j = j0 + b;
        if (tid < S) {
          // (part 0 of the plain chain: its pairs, then rule 8 on the same accumulator; then the other parts)
          double t = s_part[b * 128 + tid];
          if (v.q.unp[j - 1]) {
            const AutomatonLayout& A = v.m.lay;
            const int32_t* I = v.m.ints;
            for (int e = I[A.right_off + tid]; e < I[A.right_off + tid + 1]; ++e) {
              if (CON && !allow_right(v.m, con, v.q.L, j, tid, I[A.right_ent + 2 * e])) continue;
              t = fma(Tr.o(j - 1, I[A.right_ent + 2 * e]), lw_right(v.m, v.q, tid, I[A.right_ent + 2 * e + 1], j - 1), t);
            }
          }
          for (int k = 1; k < nparts; ++k) t += s_part[b * 128 + k * S + tid];
          v.in.o(j, tid) = t;
          if (n_ring > 0) Tr.o(j, tid) = t;
        }
        __syncthreads();
      }
    }
  } else {
  const bool ahead_on = STAGE && S <= 128 && part < nparts && !(a.dbg & 8192);
  ExtAhead nxt = ahead_on ? fetch_ahead(1, ps, part) : ExtAhead{-1, 0, 0., {0., 0., 0., 0.}, {0, 0, 0, 0}};
  for (int j = 1; j <= L; ++j) {
    if (S <= 128) {
      if (ahead_on) {
        const ExtAhead cur = nxt;
        nxt = fetch_ahead(j + 1, ps, part);          // (in flight while this step is summed and the workgroup meets)
        s_part[tid] = step_with(cur, j, ps, part, true);
      } else if (part < nparts) s_part[tid] = lin_inside_ext_part<CON>(v.m, v.q, Tr, j, ps, con, part, nparts);
      __syncthreads();
      if (tid < S) {
        double t = 0.;
        for (int k = 0; k < nparts; ++k) t += s_part[k * S + tid];
        v.in.o(j, tid) = t;
        if (n_ring > 0) Tr.o(j, tid) = t;
      }
    } else {
      for (int s = tid; s < S; s += 128) lin_inside_ext_target<CON>(v.m, v.q, v.in, j, s, con);
    }
    __syncthreads();
  }
  }
  if (tid == 0) {
    const double Zo = lin_part(v.m, v.in, true, true), Za = lin_part(v.m, v.in, true, false), Zn = lin_part(v.m, v.in, false, true);
    double sl = 0.;   // log2 of prod_p psb[base(p)]
    for (int p = 0; p < L; ++p) sl += a.lin[kLinPl2 + v.q.seq[p]];
    const double ln2 = 0.69314718055994530942;
    v.zs[0] = Zo; v.zs[1] = Za; v.zs[2] = Zn; v.zs[3] = sl;
    const SeqPlan p = a.plans[v.n];
    v.row[0] = (Zo > 0.) ? log(Zo) - sl * ln2 : ELEMDP_NEG_INF;
    v.row[1] = (Za > 0.) ? log(Za) - sl * ln2 : ELEMDP_NEG_INF;
    v.row[2] = (Zn > 0.) ? log(Zn) - sl * ln2 : ELEMDP_NEG_INF;
    if (CON && v.row[4] != 0.) {
      // (second scan pass of a sequence the first one already flagged: nothing to add)
    } else if (out_of_range(Zo) || (!a.scan && (out_of_range(Za) || out_of_range(Zn)))) {
      // outside the double range (or a structurally empty component): the log-space pipeline re-evaluates the
      // sequence and applies the reference's skip rule (motif_trainer.hpp:211-215)
      v.row[3] = 0.; v.row[4] = 2.; v.row[5] = 0.;
      const int k = atomicAdd(&a.flagged[0], 1);
      a.flagged[1 + k] = v.n;
    } else {
      // Z(ari,nasi) - Z(label); with --lik-ratio a sequence without motif contributes Z(ari) - Z(ari,nasi)
      v.row[3] = p.positive ? log1p(Zn / Za) : (a.lik_ratio ? -log1p(Zn / Za) : log1p(Za / Zn));
      v.row[4] = 0.;
      v.row[5] = p.bpp_eff;
    }
  }
}

__host__ __device__ inline int ext_stat_doubles(int nt, int det) { return (det ? 2 : 1) * (2 * nt + 4); }
struct LPass { double invZ; bool ari, nasi, skip; int en_off, eh_off; double invZs; bool merged; };
// a value that is the same in every lane, moved to scalar registers (it would otherwise occupy vector registers for the whole kernel)
__device__ __forceinline__ double uniform_f64(double x) {
  union { double d; int i[2]; } u;
  u.d = x;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
// schedule 0 (reference): pass 0 = terminals (ari,nasi), pass 1 = the label's mask; schedule 1: ari only / nasi only
__device__ __forceinline__ LPass lpass(const LinArgs& a, const LViews& v) {
  LPass pi;
  const int nt = a.lay.n_theta;
  const bool positive = v.positive;
  pi.skip = v.row[4] != 0.;
  double Z;
  if (a.schedule == 0) {
    if (a.pass == 0) { Z = v.zs[0]; pi.ari = true; pi.nasi = true; }
    else {   // (--lik-ratio: the "has motif" terminals for both labels, roles swapped in k4_combine)
      const bool use_ari = positive || a.lik_ratio;
      Z = use_ari ? v.zs[1] : v.zs[2]; pi.ari = use_ari; pi.nasi = !use_ari;
    }
  } else {
    if (a.pass == 0) { Z = v.zs[1]; pi.ari = true; pi.nasi = false; }
    else { Z = v.zs[2]; pi.ari = false; pi.nasi = true; }
  }
  // schedule 1 on an automaton with the shadow state: ONE sweep, "has motif" terminals for the pattern's states (world 0,
  // Z(ari)), the "no motif" terminal for the shadow of (0,0) (world 1, Z(nasi))
  pi.merged = a.schedule == 1 && a.lay.shadow >= 0;
  pi.invZs = uniform_f64(pi.merged ? 1. / v.zs[2] : 0.);
  pi.invZ = uniform_f64(1. / Z);
  pi.en_off = 6 + a.pass * nt;
  pi.eh_off = 6 + 2 * nt + 2 * a.pass;
  return pi;
}

// l_en: [copies][2 worlds x n_theta | 2 worlds x 2] (one copy, or one per wave in the deterministic mode: rep_sum); `slot`:
// the row of the sequence this workgroup adds to in the deterministic mode (LinArgs::det_rows)
__device__ __forceinline__ void lflush(const LinArgs& a, const LViews& v, const LPass& pi, LinSink& sink, double* l_en, int nthreads, int slot) {
  const int nt = a.lay.n_theta, ES = 2 * nt + 4;
  const int NW = a.det ? nthreads / 64 : 1;
  double* l_ehA = l_en + (a.det ? (threadIdx.x >> 6) * ES : 0) + 2 * nt;   // [worlds][2] of this lane's copy
  const int nw = pi.merged ? 2 : 1;
  const double e0 = wave_sum(sink.world == 0 ? sink.eh0 : 0.), e1 = wave_sum(sink.world == 0 ? sink.eh1 : 0.);
  if ((threadIdx.x & 63) == 0) {
    if (e0 != 0.) atomicAdd(&l_ehA[0], e0);
    if (e1 != 0.) atomicAdd(&l_ehA[1], e1);
  }
  if (pi.merged) {
    const double f0 = wave_sum(sink.world == 1 ? sink.eh0 : 0.), f1 = wave_sum(sink.world == 1 ? sink.eh1 : 0.);
    if ((threadIdx.x & 63) == 0) {
      if (f0 != 0.) atomicAdd(&l_ehA[2], f0);
      if (f1 != 0.) atomicAdd(&l_ehA[3], f1);
    }
  }
  __syncthreads();
  double* drow = a.det ? a.det_rows + ((size_t)v.n * a.det_nslot + slot) * a.out_stride : nullptr;
  for (int t = threadIdx.x; t < nw * nt + 2 * nw; t += nthreads) {
    const bool is_en = t < nw * nt;
    const int src = is_en ? t : 2 * nt + (t - nw * nt);                       // (world 1 lands on the second set: en_off + n_theta + .)
    const int dst = is_en ? pi.en_off + t : pi.eh_off + (t - nw * nt);
    const double val = rep_sum(l_en + src, NW, ES);
    if (val == 0.) continue;
    if (a.det) drow[dst] += val;      // (this workgroup's own row: launches of a stream are ordered)
    else atomicAdd(&v.row[dst], val);
  }
}

// position-posterior accumulators of the scan passes (linear, global memory, batch offsets)
template <int MODE>
__device__ __forceinline__ void scan_sink(const LinArgs& a, const LViews& v, LinSink& sink) {
  if (MODE == OUT_SCAN) { sink.pos0 = a.pos_start + v.seq_base; sink.pos1 = a.pos_inner + v.seq_base; }
  if (MODE == OUT_END) sink.pos2 = a.pos_end + v.pos_base;
}

// argmax with the reference's tie rule (the LAST maximum, util.hpp:232-241) over the linear posteriors of one sequence,
// then the logarithms the scan record prints.  WHICH 0: start (+ inner, exist prob); 1: end.
template <int WHICH>
__global__ __launch_bounds__(64) void k5_pick(LinArgs a, int G) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= G) return;
  const SeqPlan p = a.plans_slot ? a.plans_slot[g] : a.plans[a.grp[g]];
  const int n = a.plans_slot ? p.index : a.grp[g];
  const int L = p.L;
  double* arr = (WHICH == 0) ? a.pos_start + p.seq_base : a.pos_end + p.pos_base;
  const int cnt = (WHICH == 0) ? L : L + 1;
  double best = -1., tot = 0.;
  int idx = 0;
  for (int t = 0; t < cnt; ++t) {
    const double x = arr[t];
    if (best <= x) { best = x; idx = t; }
    tot += x;
    arr[t] = (x > 0.) ? log(x) : ELEMDP_NEG_INF;
  }
  if (WHICH == 0) {
    a.ys[n] = idx;
    a.exist[n] = tot;
    double* inn = a.pos_inner + p.seq_base;
    for (int t = 0; t < L; ++t) { const double x = inn[t]; inn[t] = (x > 0.) ? log(x) : ELEMDP_NEG_INF; }
  } else {
    a.ye[n] = idx;
  }
}

// ---- exterior chain of an outside pass
// (128 threads, or 512 = kExtBlock x 128: the pair terms of four consecutive steps side by side, as in k4_in_ext)
template <int MODE, bool STAGE>
__global__ __launch_bounds__(128 * kExtBlock) void k4_out_ext(LinArgs a) {
  extern __shared__ double l_stat[];   // 2 * (n_theta + 2): statistics of the two worlds, then the staged context
  __shared__ AutomatonLayout s_lay;
  const int NT = (int)blockDim.x, KB = NT / 128;
  stage_layout(a, &s_lay, NT);
  LViews v(s_lay);
  make_lviews(a, blockIdx.x, v);
  const LPass pi = lpass(a, v);
  if (pi.skip) return;
  const int n_stat = ext_stat_doubles(a.lay.n_theta, a.det);   // (a copy per wave in the deterministic mode)
  const int n_ring = (STAGE && a.ext_ring) ? ext_ring_doubles(n_stat, a.wmax, a.lay.S, kLinEth + a.lay.n_theta, a.lmax, a.nword_max, a.n_stage) : 0;
  stage_ext_context<STAGE>(a, v, reinterpret_cast<unsigned char*>(l_stat), n_stat + n_ring, NT);
  const int S = a.lay.n_active, tid = threadIdx.x, nt = a.lay.n_theta;
  TableView Or = v.out;     // the chain's rows through the LDS ring
  if (n_ring > 0) { Or.ext = l_stat + n_stat; Or.omask = (uint32_t)ext_ring_rows(a.wmax) - 1u; }
  double* l_en = l_stat + (a.det ? (tid >> 6) * (2 * nt + 4) : 0);   // the copy this lane adds to
  for (int t = tid; t < n_stat; t += NT) l_stat[t] = 0.;
  __shared__ double s_part[128 * kExtBlock];
  const int nparts = (S <= 128) ? 128 / S : 1;
  const int blk = tid >> 7, t128 = tid & 127;       // step of the block this lane sums (0 with 128 threads)
  const int part = t128 / S, ps = t128 - part * S;
  // (merged schedule: the lane of the shadow state works in world 1 -- its own Z and statistics; S <= 128 there)
  const bool shadow_lane = pi.merged && ps == a.lay.shadow;
  LinSink sink;
  sink.world = shadow_lane ? 1 : 0;
  sink.en_ = l_en + (shadow_lane ? nt : 0);
  sink.eh0 = sink.eh1 = 0.;
  scan_sink<MODE>(a, v, sink);
  LinOutCtx<LinSink> x{v.m, v.q, v.in, Or, shadow_lane ? pi.invZs : pi.invZ, sink, Constraint{MODE == OUT_END ? a.ys[v.n] : -1, -1, 0}};
  for (int s = tid; s < S; s += NT) {
    double t = 0.;
    if (pi.nasi && s == a.lay.s00) t = 1.;
    if (pi.ari && (s == a.lay.s0m1 || s == a.lay.s0m2)) t = 1.;
    if (pi.merged && s == a.lay.shadow) t = 1.;
    v.out.o(v.q.L, s) = t;
    if (n_ring > 0) Or.o(v.q.L, s) = t;
  }
  __syncthreads();
  if (KB > 1) {   // blocked chain (S <= 128, staged context, not the deterministic mode: the launcher's condition)
    for (int i0 = v.q.L - 1; i0 >= 0; i0 -= KB) {
      const int ia = i0 - blk;
      if (part < nparts && ia >= 0) s_part[tid] = lin_outside_ext_pairs<MODE>(x, ia, ps, part, nparts);
      __syncthreads();
      for (int b = 0; b < KB && i0 - b >= 0; ++b) {
        const int i = i0 - b;
        if (tid < S) {
          double t = lin_outside_ext_rule8<MODE>(x, i, tid);
          for (int k = 0; k < nparts; ++k) t += s_part[b * 128 + k * S + tid];
          v.out.o(i, tid) = t;
          if (n_ring > 0) Or.o(i, tid) = t;
        }
        __syncthreads();
      }
    }
  } else
  for (int i = v.q.L - 1; i >= 0; --i) {
    if (S <= 128) {
      if (part < nparts) s_part[tid] = lin_outside_ext_part<MODE>(x, i, ps, part, nparts);
      __syncthreads();
      if (tid < S) {
        double t = 0.;
        for (int k = 0; k < nparts; ++k) t += s_part[k * S + tid];
        v.out.o(i, tid) = t;
        if (n_ring > 0) Or.o(i, tid) = t;
      }
    } else {
      for (int s = tid; s < S; s += 128) lin_outside_ext_target<MODE>(x, i, s);
    }
    __syncthreads();
  }
  if (MODE == OUT_TRAIN || MODE == OUT_SCAN) lflush(a, v, pi, sink, l_stat, NT, a.det_nslot - 1);
}

// ---- rule 7, outside direction: P(i,j,tgt) as a child of the exterior chain = sum over the split entries of tgt of
// out O(j,par) * in O(i,s2) * exp(lambda_par e_ext(i,j)).  Both chains are complete before the band sweep starts, so the
// term of every pair cell goes into its (still unwritten) outside P entry by one throughput kernel -- all states, zeros
// included: the table slot still holds another sequence's values -- and the unary phase of k4_out adds it to its sums
// (it used to be gathered in k4_out's pair phase: 7 ms of the 145 ms evaluation of 4096 sequences).
__global__ __launch_bounds__(kThreads) void k4_r7(LinArgs a) {
  __shared__ AutomatonLayout s_lay;
  stage_layout(a, &s_lay, kThreads);
  LViews v(s_lay);
  make_lviews(a, blockIdx.y, v);
  __syncthreads();
  const AutomatonLayout& A = s_lay;
  const int L = v.q.L, W = v.q.W, S = A.S;
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= (L + 1) * (W + 1)) return;
  const int i = c / (W + 1), d = c - i * (W + 1);
  if (!v.q.pair_ok(i, d)) return;
  const int j = i + d;
  const int32_t* G = v.m.big;
  const double x0 = xw_cell(v.q, 0, XT_EXT, v.q.cell(i, d)), x1 = xw_cell(v.q, 1, XT_EXT, v.q.cell(i, d));
  for (int s = 0; s < S; ++s) {
    double acc = 0.;
    for (int u = G[A.split2_off + s]; u < G[A.split2_off + s + 1]; ++u) {
      const int par = G[A.split2_ent + 2 * u], s2i = G[A.split2_ent + 2 * u + 1];
      acc = fma(v.out.o(j, par), v.in.o(i, s2i) * (lamk(v.m, par) ? x1 : x0), acc);
    }
    v.out.st(ST_P, d, i, s, acc);
  }
}

// ---- outside, diagonal d: dynamic LDS = 4 * cpb * S + n_theta + 2 doubles
// W6: asked for six waves per SIMD (80 registers, a few spilled dwords) -- taken by the launcher where six workgroups fit the LDS
template <int MODE, bool BIG, bool FAST = false, int FP = kFastP, bool W6 = false, bool MB = false>
__global__ __launch_bounds__(kBT, W6 ? ELEMDP_LB_OUT6 : ELEMDP_LB_OUT) void k4_out(LinArgs a) {
  extern __shared__ double lds[];
  // (the automaton layout is read from the kernel arguments: constant offsets, scalar registers)
  PhaseClock pc;
  pc.start(a.prof);

  unsigned bx, by;
  swizzled_block(bx, by);
  LViews v(a.lay);
  make_lviews(a, by, v);
  const LPass pi = lpass(a, v);
  const AutomatonLayout& A = a.lay;
  const int S = a.lay.S, NA = a.lay.n_active, d = a.d, cpb = a.cpb, tid = threadIdx.x, nt = a.lay.n_theta;
  // (nblk blocks of cpb cells per workgroup, context staged once: see k4_in)
  const int nblk = (MB && a.nblk > 1) ? a.nblk : 1, cpbT = cpb * nblk;
  if (d > v.q.W) return;
  const int L = v.q.L, W = v.q.W;
  const int ncell = L - d + 1, i0T = bx * cpbT;
  if (i0T >= ncell) return;
  const int ncT = (cpbT < ncell - i0T) ? cpbT : ncell - i0T;
  if (MODE == OUT_END) {
    // Under the start constraint the motif begins at Ys: a cell that ends at or before Ys holds no part of it, so no transition in
    // it can be an end of the motif (its posterior is an exact 0: a derivation whose motif began earlier emits Ys with weight 0),
    // and nothing that is swept reads its outside value -- parents, item sums and pair entries all look at cells that contain the
    // reader.  Workgroups whose cells all satisfy j <= Ys return: on average half of the sweep.
    if (!(a.dbg & 4096) && i0T + ncT - 1 + d <= a.ys[v.n]) return;
  }
  const int HD = FAST ? A.n_lane : S;   // (as in k4_in)
  const int CS = cpb * HD;
  // ONE copy of the heavy sums (deterministic mode: each gets its adds from one wave, see the top of the file); the statistics,
  // which every lane may add to, one copy per wave there (rep_sum)
  constexpr int NW = 1;
  const int NWS = a.det ? kBT / 64 : 1, wvd = a.det ? tid >> 6 : 0;
  const int HS = 4 * CS, ES = 2 * nt + 4;      // doubles of the heavy sums / of one copy of the statistics
  double* h1 = lds;
  double* h2 = h1 + CS;
  double* hp = h2 + CS;
  double* hl = hp + CS;
  double* h1A = h1;
  double* h2A = h2;
  double* hpA = hp;
  double* l_en0 = lds + HS;                    // statistics: [copies][2 worlds x n_theta | 2 worlds x 2]
  double* l_en = l_en0 + wvd * ES;             // ... this lane's copy
  double* l_eh = l_en + 2 * nt;
  double* l_pos = l_en0 + NWS * ES;            // scan: [2][win] position posteriors of the window (start, inner | end, -)
  const int win = cpbT + a.wmax + 3;
  constexpr bool kScanMode = MODE == OUT_SCAN || MODE == OUT_END;
  double* sOB1 = l_pos + (kScanMode ? 2 * win : 0);   // item records of the three roles (kRecOut doubles; no position window in training)
  const BlockLds BL = block_lds(out_doubles(CS, nt, win, NWS, kScanMode), cpbT, a.n_lin, win, FAST ? a.lay.fb_out_n : staged_ints(a.lay, a.n_stage, 1), 3 * cpbT, FAST ? kCellOutD : 0);
  // cell records of the table-driven unary phase (see k4_in): twelve global values per cell, fetched with the context
  constexpr int kCRout = (ELEMDP_CPB_MAX * 12 + kBT - 1) / kBT;
  double crx[kCRout];
  if (FAST) {
#pragma unroll
    for (int r = 0; r < kCRout; ++r) {
      crx[r] = 0.;
      if (r * kBT < cpbT * 12) {   // (uniform: most automata need the first round only)
        const int t = tid + r * kBT, c0 = t / 12, c = c0 < ncT ? c0 : 0;
        crx[r] = cell_out_fetch(v.q, d, i0T + c, t - c0 * 12);
      }
    }
  }
  if (a.dbg & 1024) { if (ncT == 12345) lds[tid] = crx[0] + pi.invZ; return; }   // (timing experiments, as in k4_in)
  const BlockCtx cx = stage_context<BIG, 1, FAST>(a, v, reinterpret_cast<unsigned char*>(lds), BL, i0T, ncT, d, cpbT);
  if (pi.skip) return;   // (tested here: the loads behind `pi` travel with those of the context instead of before them)
  int* dmT = cx.dm; int* cntsT = cx.cnts; int* pre = cx.pre; int* baseT = cx.base;
  const int32_t* G = v.m.big;
  double* crecT = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(lds) + BL.crec);
  int* crflT = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(lds) + BL.crfl);
  const int n_zero = HS + NWS * ES + ((MODE == OUT_SCAN || MODE == OUT_END) ? 2 * win : 0);
  for (int t = tid; t < n_zero; t += kBT) lds[t] = 0.;
  __syncthreads();
  if (a.dbg & 2048) return;
  pc.mark<5>();
  if (FAST) {   // (read by the unary phase, behind the barriers of the heavy sums)
#pragma unroll
    for (int r = 0; r < kCRout; ++r) {
      const int t = tid + r * kBT, c = t / 12, k = t - c * 12;
      if (c < ncT) {
        const int i = i0T + c;
        const bool on = k < 4 ? v.q.e_ok(i, d) : k < 6 ? v.q.pair_ok(i, d) : k < 8 ? (v.q.pair_ok(i - 1, d + 2) && v.q.pair_ok(i, d)) : true;
        crecT[c * kCellOutD + 2 + k] = on ? crx[r] : 0.;
      }
    }
    for (int c = tid; c < ncT; c += kBT) {
      const int i = i0T + c, j = i + d;
      int fl = cell_out_flags(v.m, v.q, d, i);
      if (MODE == OUT_END) { const int ys = a.ys[v.n]; fl |= (i - 1 == ys ? CF_YL : 0) | (j == ys ? CF_YR : 0) | (L == j + 1 ? CF_JLAST : 0); }
      crflT[c] = fl;
      crecT[c * kCellOutD] = v.q.ews[i > 0 ? i - 1 : 0];
      crecT[c * kCellOutD + 1] = v.q.ews[j < L ? j : L];
    }
  }
  LinSink sink;
  sink.en_ = l_en;
  sink.eh0 = sink.eh1 = 0.;
  // scan: the position posteriors of this workgroup-diagonal are summed in LDS over the window of positions it touches
  // ([p0, p0 + win): i0-1 .. i0+nc+d, the window of the staged context) and added to the sequence's arrays once at the end
  const int pos_p0 = (i0T > 0) ? i0T - 1 : 0;
  if (MODE == OUT_SCAN) { sink.pos0 = l_pos - pos_p0; sink.pos1 = l_pos + win - pos_p0; }
  if (MODE == OUT_END) sink.pos2 = l_pos - pos_p0;
  const TableView& in = v.in;
  const TableView& out = v.out;
  const int nq = ((a.dbg & 2) || (ELEMDP_KO & 2)) ? 0 : A.n_quad;
  const double* IB = in.band;
  const double* OB = out.band;
  // CSR ranges of the item sums of the three roles of all cells of the workgroup, [block][role][cpb] (consumed behind the first
  // pair phase, whose loads they travel with; the ranges of cells past the end are empty)
  for (int vc = tid; vc < 3 * cpbT; vc += kBT) {
    const int kb = div_rcp(vc, a.rcp_3cpb), vr = vc - kb * 3 * cpb;
    const int role = div_rcp(vr, a.rcp_cpb), cT = kb * cpb + (vr - role * cpb);
    const bool have = cT < ncT;
    const int i = i0T + (have ? cT : 0);
    const int cell = v.q.cell(i, d);
    const int32_t* off = role == 0 ? v.q.by_inner_off : role == 1 ? v.q.by_left_off : v.q.by_right_off;
    int n0 = 0, n1 = 0;
    // (table-driven kernels: no loop sums on the diagonal d = 0 -- the outside value of an EMPTY loop L(i,i) has no reader: it has
    // no children, and the statistics of the emissions into it use the parent's value.  Those cells own the records of every
    // stack and bulge of the sequence.  The generic kernels, whose tables debug_tables exports, keep them.)
    if (have && nq > 0 && (role != 0 || v.q.pair_ok(i, d)) && !(FAST && role != 0 && d == 0)) { n0 = off[cell]; n1 = off[cell + 1]; }
    baseT[vc] = n0;
    cntsT[vc] = (n1 > n0) ? n1 - n0 : 0;
  }
  const int nv = 3 * cpb;
  const int tid_wg = tid;
  for (int blk = 0; MB ? blk * cpb < ncT : blk < 1; ++blk) {
  int tid = tid_wg;   // (behind an empty asm: see k4_in)
  if (MB) asm volatile("" : "+v"(tid));
  const int i0 = i0T + blk * cpb, nc = (cpb < ncT - blk * cpb) ? cpb : ncT - blk * cpb;
  if (MODE == OUT_END && !(a.dbg & 4096) && i0 + nc - 1 + d <= a.ys[v.n]) continue;   // (uniform; see the workgroup test above)
  const int* dm = dmT + blk * cpb;
  const int* cnts = cntsT + blk * 3 * cpb;
  const int* base = baseT + blk * 3 * cpb;
  const double* crec = crecT + blk * cpb * kCellOutD;
  const int* crfl = crflT + blk * cpb;
  // rule 2, factorised, outside direction (lin_rules.h: lheavy_o1 / lheavy_o2):
  //   h1[c][s1] = H1 = sum over the stems (j, l) that start at the cell's end j = i + d:  outA(i,l,p) * P(j,l,t) * xml(j,l)
  //   h2[c][t]  = HA = sum_{ii < i} outA(ii,j,p) * 1(ii,i,s1), only where the cell itself is a stem P(i,j)
  // lanes = (cell, pair); stems four at a time.  Both read outA of larger spans only (earlier launches).
  {
    const int nA = A.n_ap;
    const int32_t* I = v.m.ints;
    const int W1 = W + 1;
    const int nwork = ((a.dbg & 1) || (ELEMDP_KO & 1)) ? 0 : nc * nA;
    // attributes of pair p: from its record in the fast blob (AutomatonLayout::fpr_out), or from the generic lists
    auto pr_s1 = [&](int p) { return FAST ? (G[A.fpr_out + 8 * p + 1] & 0xff) : I[A.ap_s1 + p]; };
    auto pr_t = [&](int p) { return FAST ? ((G[A.fpr_out + 8 * p + 1] >> 8) & 0xff) : I[A.ap_t + p]; };
    auto pr_c1 = [&](int p) { return FAST ? fcol(G[A.fpr_out + 8 * p], 0) : in.col(ST_1, I[A.ap_s1 + p]); };
    auto pr_cP = [&](int p) { return FAST ? fcol(G[A.fpr_out + 8 * p], 1) : in.col(ST_P, I[A.ap_t + p]); };
    auto pr_kl = [&](int p) { return FAST ? ((G[A.fpr_out + 8 * p] >> 24) & 1) : lamk(v.m, I[A.ap_t + p]); };
    // HA: the few stem cells of the workgroup, work item = (stem cell, parent row ii = i - b, pair) over ALL of them as one
    // flat list, four work items' loads in flight per lane.  The loads of the first round are issued here, ahead of the H1
    // sums: both read outA of larger spans only, so one round trip serves both.  (The rule-7 term of the stem cells' P states
    // is already in the table: k4_r7.)
    unsigned long long stems = 0;             // bit c: cell c of the workgroup is a pair (cpb <= 64)
    for (int c = 0; c < nc; ++c) stems |= v.q.pair_ok(i0 + c, d) ? (1ull << c) : 0ull;
    if ((a.dbg & 1) || (ELEMDP_KO & 1)) stems = 0;
    const int nb = W - d;                     // b = 1 .. nb: parent span d + b <= W
    const int per = nb * nA, total = __popcll(stems) * per;
    constexpr int kHA = 4;
    double ha_oa[kHA], ha_x1[kHA];
    int ha_idx[kHA];
    // work item (sc-th stem cell, r = (b - 1) * nA + p) into slot u of the batch
    auto ha_slot = [&](int u, bool valid, int sc, int r) {
      {
        unsigned long long m = stems;
        for (int k = 0; k < sc; ++k) m &= m - 1;          // the sc-th stem cell
        const int c = stems ? __builtin_ctzll(m) : 0;
        const int b = 1 + div_rcp(r, a.rcp_nap), p = r - (b - 1) * nA;
        const int ii = i0 + c - b;
        const int dmii = (valid && ii >= 0) ? (int)v.q.dmin[ii] : 0;
        const bool ok = dmii > 0 && b >= dmii;              // 1(ii, i, .) is parsable (then the pair entries of (ii, d + b) exist)
        ha_oa[u] = out.lda(d + b, ii, p, ok);
        ha_x1[u] = in.ldc(ST_1, b, ii, pr_c1(p), ok);
        ha_idx[u] = c * HD + pr_t(p);
      }
    };
    auto ha_load = [&](int w0) {      // flat list over all stem cells of the block: w = sc * per + r
#pragma unroll
      for (int u = 0; u < kHA; ++u) {
        const int w = w0 + u * kBT;
        const bool valid = w < total;
        const int sc = valid ? div_small(w, per) : 0;
        ha_slot(u, valid, sc, valid ? w - sc * per : 0);
      }
    };
    auto ha_add = [&]() {
#pragma unroll
      for (int u = 0; u < kHA; ++u) {
        const double term = ha_oa[u] * ha_x1[u];
        if (term != 0.) atomicAdd(&h2A[ha_idx[u]], term);
      }
    };
    // (unconditional, like the add below: a value that is defined under a condition inside the block loop counts as live around
    // the whole loop -- 16 registers here)
    ha_load(a.det ? total : tid);      // (deterministic mode: HA follows the H1 sums, a stem cell per wave)
    // (deterministic mode: a cell's pairs in one wave, as in k4_in)
    const int pad = (a.det && a.det_sh >= 0) ? (1 << a.det_sh) : nA;
    const bool one_wave = a.det && a.det_sh < 0;
    const int nwork_l = nwork ? nc * pad : 0;
    for (int w = one_wave ? ((tid < 64) ? tid : nwork_l) : tid; w < nwork_l; w += one_wave ? 64 : kBT) {
      const int c = (a.det && a.det_sh >= 0) ? (w >> a.det_sh) : div_rcp(w, a.rcp_nap), p = w - c * pad;
      if (p >= nA) continue;
      const int i = i0 + c, j = i + d;
      const int s1 = pr_s1(p);
      const int cP = pr_cP(p);
      const int dmi = dm[c];
      if (dmi > 0 && dmi <= d) {     // left_ok(i, d)  (a dead child 1(i,j,s1) drops the sum in the unary phase)
        const int hi = (W - d < L - j) ? W - d : L - j;
        const double* xml = v.q.xwc + (size_t)(pr_kl(p) * 5 + XT_ML) * v.q.xwc_stride;
        double acc = 0.;
        BitIter it;
        it.init(v.q.okbits, j * W1, 1, hi);
        if (FAST) {   // (b = X = P * exp(lambda e_ml) from the B plane's rows, see fast_inside_unary; no weight load)
          for (;;) {
            int sp[kStems];
            sp[0] = it.next();
            if (sp[0] < 0) break;
#pragma unroll
            for (int u = 1; u < kStems; ++u) sp[u] = sp[u - 1] < 0 ? -1 : it.next();
            double sa[kStems], sb[kStems];
#pragma unroll
            for (int u = 0; u < kStems; ++u) {
              const bool g = sp[u] >= 0;
              const int q = g ? sp[u] : 1;
              sa[u] = out.lda(d + q, i, p, g);      // (the pair entries of (i, d + sp) exist: dmin[i] <= d < d + sp)
              sb[u] = in.ldc(ST_B, q, j, cP, g);
            }
#pragma unroll
            for (int u = 0; u < kStems; ++u) acc = fma(sa[u], sb[u], acc);
            if (sp[kStems - 1] < 0) break;
          }
        } else
        for (;;) {
          const int sp0 = it.next();
          if (sp0 < 0) break;
          const int sp1 = it.next(), sp2 = (sp1 < 0) ? -1 : it.next(), sp3 = (sp2 < 0) ? -1 : it.next();
          const int q1 = sp1 < 0 ? sp0 : sp1, q2 = sp2 < 0 ? sp0 : sp2, q3 = sp3 < 0 ? sp0 : sp3;
          // (the pair entries of (i, d + sp) exist: dmin[i] <= d < d + sp; the stems are kept pairs)
          const double a0 = out.a(d + sp0, i, p), b0 = in.ldc(ST_P, sp0, j, cP), c0 = xml[v.q.cell(j, sp0)];
          const double a1 = out.a(d + q1, i, p), b1 = in.ldc(ST_P, q1, j, cP), c1 = xml[v.q.cell(j, q1)];
          const double a2 = out.a(d + q2, i, p), b2 = in.ldc(ST_P, q2, j, cP), c2 = xml[v.q.cell(j, q2)];
          const double a3 = out.a(d + q3, i, p), b3 = in.ldc(ST_P, q3, j, cP), c3 = xml[v.q.cell(j, q3)];
          acc = fma(a0, b0 * c0, acc);
          if (sp1 >= 0) acc = fma(a1, b1 * c1, acc);
          if (sp2 >= 0) acc = fma(a2, b2 * c2, acc);
          if (sp3 >= 0) acc = fma(a3, b3 * c3, acc);
          if (sp3 < 0) break;
        }
        if (acc != 0.) atomicAdd(&h1A[c * HD + s1], acc);
      }
    }
    if (!a.det) {
      ha_add();
      for (int w0 = tid + kHA * kBT; w0 < total; w0 += kHA * kBT) { ha_load(w0); ha_add(); }
    } else {      // every stem cell's sums from ONE wave: wave w takes the stem cells w, w + 4, ..; its lanes the (row, pair) items
      const int nst = __popcll(stems), wvh = tid >> 6, lnh = tid & 63;
      for (int sc = wvh; sc < nst; sc += kWaves)
        for (int r0 = lnh; r0 < per; r0 += 64 * kHA) {
#pragma unroll
          for (int u = 0; u < kHA; ++u) { const int r = r0 + 64 * u; ha_slot(u, r < per, sc, r < per ? r : 0); }
          ha_add();
        }
    }
  }
  __syncthreads();
  pc.mark<6>();
  // HP / HL: the three roles of a cell in the interior loops around it (rule 6c) -- inner pair P(i,j,tgt) of E(it.i,it.j,par)
  // (with the energy statistic of the rule), left loop L(i,j,tgt) = L(it.i,it.k), right loop L(i,j,tgt) = L(it.l,it.j) --
  // as ONE flat list of work items (role, item, tuple): CSR range per (role, cell) -> LDS prefix; the item records and
  // their weights are staged into the (now free) operand staging area by all lanes with one round of loads; a work item
  // then needs a single round of table loads, selected by role without branches.
  {
    // (the unary phase of the previous block left out B of its targets in `hp`, which the pair entries behind it have read)
    if (MB && blk > 0) for (int t = tid; t < CS; t += kBT) hp[t] = 0.;
    lds_prefix(nv, tid, cnts, pre);
    __syncthreads();
    const int n_rec = pre[nv];
    // record area: LoopItem it[cap], double xw[2][cap], int meta[cap] (role << 16 | cell)
    const int cap = (kRecOut * 8) / 40;
    LoopItem* r_it = reinterpret_cast<LoopItem*>(sOB1);
    double* r_xw = reinterpret_cast<double*>(r_it + cap);
    int* r_meta = reinterpret_cast<int*>(r_xw + 2 * cap);
    // energy statistics of rule 6c: a lane serves both worlds here, so [world][stack / other] sums of their own, added to
    // the workgroup's before the unary phase (four accumulators less across it)
    double ew[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ew[k] = 0.; asm volatile("" : "+v"(ew[k])); }   // (pinned here: the zeros would be set up at the top of the block loop)
    for (int p0 = 0; p0 < n_rec; p0 += cap) {
      const int np = (cap < n_rec - p0) ? cap : n_rec - p0;
      for (int x = tid; x < np; x += kBT) {
        const int p = p0 + x;
        int lo, n;
        outer_locate(p, nv, pre, base, lo, n);   // the (role, cell) owning record p
        const int role = (lo >= cpb) + (lo >= 2 * cpb);
        const LoopItem* src = role == 0 ? v.q.items_inner : role == 1 ? v.q.items_left : v.q.items_right;
        const LoopItem itv = src[n];
        const int meta = (role << 16) | (lo - role * cpb);
        r_it[x] = itv;
        r_xw[x] = lin_weight(v.m.lambda[0], itv.tsc);          // (exp(lambda_k * tsc), as in outer_stage)
        r_xw[cap + x] = lin_weight(v.m.lambda[1], itv.tsc);
        r_meta[x] = meta;
      }
      __syncthreads();
      pc.mark<8>();
      // lane = item record of any role (the column records of the three roles' tuple lists lie at qc_out1 + role * 2 * nq).
      // Operand liveness by construction: the outer cell of an item is a parsable E cell, its inner pair a kept pair, the loops L
      // are stored everywhere; role 0 only has records at pair cells.
      const int wv = tid >> 6, lane = tid & 63;
      constexpr int kTU = kTUout;
      for (int xb = 0; xb < np; xb += 64) {
        const int x = xb + lane;
        const bool valid = x < np;
        const int xc = valid ? x : np - 1;
        const LoopItem it = r_it[xc];
        const int role = r_meta[xc] >> 16, c = r_meta[xc] & 0xffff;
        const int i = i0 + c, j = i + d;
        const uint32_t rE = out.cidx(ST_E, it.j - it.i, it.i, 0);
        const uint32_t rPi = in.cidx(ST_P, it.l - it.k, it.k, 0);
        const uint32_t r1 = role == 0 ? in.cidx(ST_L, i - it.i, it.i, 0) : rPi;
        const uint32_t r2 = role == 0 ? in.cidx(ST_L, it.j - j, j, 0) : role == 1 ? in.cidx(ST_L, it.j - it.l, it.l, 0) : in.cidx(ST_L, it.k - it.i, it.i, 0);
        const uint32_t rA = role == 0 ? in.cidx(ST_P, d, i, 0) : in.cidx(ST_L, d, i, 0);
        const double xw0 = r_xw[xc], xw1 = r_xw[cap + xc];
        double* hrow = hpA + (role == 0 ? 0 : CS) + c * HD;   // hp, or hl = hp + CS
        const int qc0 = (FAST ? A.fqc_out : A.qc_out1) + role * 2 * nq;
        // the wave's tuples of the lane's role: every fourth one, or (deterministic mode) the ones whose targets are the wave's own
        // (qd_out: offsets + ids per role); the loop runs to the longest of the three roles' shares
        const int qd0 = (FAST ? A.fqd_out : A.qd_out) + role * (5 + nq);
        const int m0 = a.det ? G[qd0 + wv] : 0;
        const int n_mine = a.det ? G[qd0 + wv + 1] - m0 : (nq > wv ? (nq - wv + kWaves - 1) / kWaves : 0);
        int n_loop = n_mine;
        if (a.det) {
          const int qdb = FAST ? A.fqd_out : A.qd_out;
          n_loop = 0;
#pragma unroll
          for (int r = 0; r < 3; ++r) { const int nr = G[qdb + r * (5 + nq) + wv + 1] - G[qdb + r * (5 + nq) + wv]; n_loop = nr > n_loop ? nr : n_loop; }
        }
        for (int k0 = 0; k0 < n_loop; k0 += kTU) {
          int qa[kTU], qb[kTU];
          double x0[kTU], x1[kTU], x2[kTU], aux[kTU];
#pragma unroll
          for (int u = 0; u < kTU; ++u) {
            const bool on = k0 + u < n_mine;
            const int kk = on ? k0 + u : 0;
            const int t = !on ? (nq > wv ? wv : 0) : a.det ? G[qd0 + 5 + m0 + kk] : wv + kWaves * kk;
            qa[u] = G[qc0 + 2 * t];
            qb[u] = on ? G[qc0 + 2 * t + 1] : (4 << 16);
            x0[u] = OB[rE + (qa[u] & 0xff)]; x1[u] = IB[r1 + ((qa[u] >> 8) & 0xff)]; x2[u] = IB[r2 + ((qa[u] >> 16) & 0xff)];
            // (own inside value of the target: the posterior of the energy statistic for an inner pair, and a reason to skip the
            // add when it is 0 -- for the loops, whose inside values are practically never 0, the load would only cost: the
            // unary phase ignores the sum of a target whose inside value is 0 anyway)
            aux[u] = role == 0 ? IB[rA + ((qa[u] >> 24) & 0xff)] : 1.;
          }
#pragma unroll
          for (int u = 0; u < kTU; ++u) {
            const double term = x0[u] * (x1[u] * x2[u]) * ((qb[u] & (1 << 16)) ? xw1 : xw0);
            if ((qb[u] & (4 << 16)) || !valid || aux[u] == 0. || term == 0.) continue;
            atomicAdd(&hrow[qb[u] & 0xffff], term);
            if (MODE == OUT_TRAIN && role == 0) {   // energy statistic of the rule (world and lambda class of the tuple's parent)
              const bool w1 = pi.merged && (qb[u] & (2 << 16));
              const bool k1 = !v.m.lam_same && (qb[u] & (1 << 16));
              const double w = it.tsc * term * (aux[u] * (w1 ? pi.invZs : pi.invZ));
              ew[0] += (!w1 && !k1) ? w : 0.;
              ew[1] += (!w1 && k1) ? w : 0.;
              ew[2] += (w1 && !k1) ? w : 0.;
              ew[3] += (w1 && k1) ? w : 0.;
            }
          }
        }
      }
      __syncthreads();
      pc.mark<9>();
    }
    if (MODE == OUT_TRAIN && n_rec > 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double e = wave_sum(ew[k]);
        if ((tid & 63) == 0 && e != 0.) atomicAdd(&l_eh[k], e);
      }
    }
  }
  pc.mark<10>();
  const int NL = FAST ? A.n_lane : NA;   // lanes per cell of the unary phase: the states that have a column at all
  // (OUT_END: the cells of this workgroup that end at or before Ys take no part -- see the workgroup test above; their parents may
  // lie in a workgroup that returned, so neither their values nor their statistics mean anything)
  const int end_ys = (MODE == OUT_END && !(a.dbg & 4096)) ? a.ys[v.n] : -1;
  // out B of a target goes to its slot of `hp` (copy 0) for the pair entries below; the lane clears the other sums it owns, for
  // the next block of the workgroup: the pair phase of that block adds to h1 / h2 while the pair entries of this one still read hp,
  // which is cleared behind the next pair phase (table-driven: HD = NL, and cells past nc take no adds)
  if (tid < nc * NL && !(a.dbg & 4) && !(ELEMDP_KO & 4)) {
    const int c = div_rcp(tid, a.rcp_lane);
    const int s = FAST ? G[A.f_live_out + tid - c * NL] : tid - c * NL;
    const int slot = FAST ? tid : c * HD + s;
    double oB = 0.;
    if (i0 + c + d > end_ys) {
      const bool w1 = pi.merged && s == A.shadow;      // the shadow state: world 1 (its own Z, second set of statistics)
      sink.world = w1 ? 1 : 0;
      sink.en_ = l_en + (w1 ? nt : 0);
      if (FAST) {
        oB = fast_outside_unary<kFastR, FP, kFastL, MODE>(A, G + A.fp_out + s * kFastW, G, v.m.lin, in, out, crec + c * kCellOutD, crfl[c],
                                                          d, i0 + c, w1 ? pi.invZs : pi.invZ, v.m.lam_same != 0, v.m.no_prf != 0, sink,
                                                          h1 + slot, CS, NW, HS, G + A.fs_out);
      } else {
        LinOutCtx<LinSink> x{v.m, v.q, in, out, w1 ? pi.invZs : pi.invZ, sink, Constraint{MODE == OUT_END ? a.ys[v.n] : -1, -1, 0}};
        HeavyOut H;
        H.H1 = rep_sum(h1 + slot, NW, HS); H.H2 = rep_sum(h2 + slot, NW, HS);
        H.HP = rep_sum(hp + slot, NW, HS); H.HL = rep_sum(hl + slot, NW, HS);
        if (!(a.dbg & 32)) H.HP += out.ld(ST_P, d, i0 + c, s, v.q.pair_ok(i0 + c, d));   // rule-7 term (k4_r7)
        H.ext_in_hp = true;
        oB = lin_outside_target_u<MODE>(x, d, i0 + c, s, H);     // out B(i,d,s)
      }
    }
    if (nblk > 1)
      for (int r = 0; r < NW; ++r) { h1[r * HS + slot] = 0.; h2[r * HS + slot] = 0.; hl[r * HS + slot] = 0.; if (r) hp[r * HS + slot] = 0.; }
    hp[slot] = oB;
  }
  __syncthreads();
  // outside values of the pair entries of the cells (lin_outside_apair): out B of the target + the tail step from
  // (i, d+1), with the statistics of the tail emissions
  if (!(a.dbg & 1) && !(a.dbg & 128) && !(ELEMDP_KO & 8)) {
    const int nA = A.n_ap;
    double* const en_keep = sink.en_;
    for (int w = tid; w < nc * nA; w += kBT) {
      const int c = div_rcp(w, a.rcp_nap), p = w - c * nA;
      if (i0 + c + d <= end_ys) continue;
      if (FAST) {   // lin_outside_apair from the pair record (AutomatonLayout::fpr_out) and the weight tables
        const int32_t* PR = G + A.fpr_out + 8 * p;
        const int r0 = PR[0], r1 = PR[1];
        const int tg = (r0 >> 16) & 0xff;
        const bool w1 = pi.merged && (r0 & (4 << 24));
        const int i = i0 + c, j = i + d, dmi = dm[c];
        if (dmi > 0 && dmi < d) {      // (otherwise the entry does not exist)
          const bool step = d + 1 <= W && j < L && v.q.unp[j];
          const int nr = step ? (r1 >> 20) & 15 : 0;
          const double a_in = in.a(d, i, p);
          int ce[kFastR];
          double op[kFastR];
#pragma unroll
          for (int u = 0; u < kFastR; ++u) {
            ce[u] = PR[5 + u];
            op[u] = out.lda(d + 1, i, u < nr ? ce[u] & 0xff : p, u < nr);
          }
          double acc = 0.;
          if (a_in != 0.) {
            acc = tg != 0xff ? hp[c * HD + tg] : 0.;
            const double inz = a_in * (w1 ? pi.invZs : pi.invZ);
            const int bj = step ? (int)v.q.seq[j] : 0;
            const double ewj = step ? v.q.ews[j] : 1.;
            double* en = l_en + (w1 ? nt : 0);
#pragma unroll
            for (int u = 0; u < kFastR; ++u)
              if (u < nr) {
                const int id = (ce[u] >> 8) & 0x7fff;
                const double term = op[u] * (v.m.lin[A.lin_wr + 5 * id + bj] * ((G[A.fe_r + 2 * id + 1] & 1) ? ewj : 1.));
                const double z = term * inz;
                if ((MODE == OUT_SCAN || MODE == OUT_END) &&
                    !fast_scan_stat<MODE, false, true>(sink, G[A.fs_out + id], i - 1, j, false, MODE == OUT_END && j == a.ys[v.n], L == j + 1, z))
                  continue;
                if (MODE != OUT_END && !v.m.no_prf && z != 0. && bj) atomicAdd(&en[G[A.fe_r + 2 * id] + bj], z);
                acc += term;
              }
          }
          out.a(d, i, p) = acc;
        }
        continue;
      }
      const int tgt = v.m.ints[A.ap_tgt + p];
      const bool w1 = pi.merged && v.m.ints[A.ap_t + p] == A.shadow;
      sink.en_ = l_en + (w1 ? nt : 0);
      LinOutCtx<LinSink> x{v.m, v.q, in, out, w1 ? pi.invZs : pi.invZ, sink, Constraint{MODE == OUT_END ? a.ys[v.n] : -1, -1, 0}};
      lin_outside_apair<MODE>(x, d, i0 + c, p, tgt >= 0 ? hp[c * HD + tgt] : 0.);
    }
    sink.en_ = en_keep;
  }
  pc.mark<11>();
  if (MB && (blk + 1) * cpb < ncT && (!FAST || (a.dbg & 4) || (ELEMDP_KO & 4))) {   // (generic rule code: the lanes do not cover every slot)
    __syncthreads();
    for (int t = tid; t < NW * HS; t += kBT) lds[t] = 0.;
    __syncthreads();
  }
  }   // blocks of the workgroup
  if (MODE == OUT_SCAN || MODE == OUT_END) {
    __syncthreads();
    const int p1 = (i0T + ncT + d < L) ? i0T + ncT + d : L;   // inclusive
    for (int t = tid; t <= p1 - pos_p0; t += kBT) {
      const double v0 = l_pos[t], v1 = l_pos[win + t];
      if (MODE == OUT_SCAN) {
        if (v0 != 0. && pos_p0 + t < L) atomicAdd(&a.pos_start[v.seq_base + pos_p0 + t], v0);
        if (v1 != 0. && pos_p0 + t < L) atomicAdd(&a.pos_inner[v.seq_base + pos_p0 + t], v1);
      } else if (v0 != 0.) {
        atomicAdd(&a.pos_end[v.pos_base + pos_p0 + t], v0);
      }
    }
  }
  if (MODE == OUT_TRAIN || MODE == OUT_SCAN) lflush(a, v, pi, sink, l_en0, kBT, (int)bx);
  pc.mark<12>();
  pc.finish();
}

// Final statistics of a sequence from those of its two outside passes (see k3_combine in train_kernels.hip):
// schedule 1: A = ari-only, B = nasi-only, F = p_a A + p_n B; (o, x) = (F, A) with motif, (F, B) without, (A, F) without
// under --lik-ratio.  schedule 0 + --lik-ratio: swap (o, x) of the sequences without motif.
__global__ __launch_bounds__(kThreads) void k4_combine(LinArgs a, int G) {
  const int g = blockIdx.x;
  if (g >= G) return;
  const int n = a.grp[g];
  double* row = a.seq_out + (size_t)n * a.out_stride;
  if (row[4] != 0.) return;
  const double* zs = a.zs + (size_t)g * 4;
  const int nt = a.lay.n_theta;
  if (a.det) {   // deterministic mode: the counts of the workgroups' own rows, in block order
    const double* dr = a.det_rows + (size_t)n * a.det_nslot * a.out_stride;
    for (int t = 6 + threadIdx.x; t < 6 + 2 * nt + 4; t += kThreads) {
      double acc = 0.;
      for (int k = 0; k < a.det_nslot; ++k) acc += dr[(size_t)k * a.out_stride + t];
      row[t] = acc;
    }
    __syncthreads();
  }
  const bool positive = a.plans[n].positive != 0;
  const double pa = zs[1] / zs[0], pn = zs[2] / zs[0];
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

// ---- Viterbi pass of the scan on the batch pipeline (log space, max-plus; rules: scan_rules.h, reference
// RNAelemScanDP::CYKFun motif_scanner.hpp:802-913).  Ties are broken by the first strictly greater candidate (:821), and
// with uniform theta ties are frequent -- so a parallel maximum has to reproduce the reference's candidate ORDER.  Every
// candidate of the two heavy lists gets its ordinal in that order (rule 2: split point, then tuple; rule 6c: item, then
// tuple); the workgroup reduces (value, ordinal) pairs: an LDS atomic max over order-preserving integer keys of the
// values, then an atomic min of the ordinals among the candidates that reached the maximum.  Values are single sums in
// the reference's association, so they are bitwise those of the serial evaluation.  No exp / log here.
__device__ __forceinline__ unsigned long long cyk_key(double v) {
  const long long b = __double_as_longlong(v + 0.);   // (-0 -> +0: equal values must have equal keys)
  return (unsigned long long)(b >= 0 ? b : (b ^ 0x7fffffffffffffffLL)) ^ 0x8000000000000000ULL;
}
__device__ __forceinline__ double cyk_unkey(unsigned long long k) {
  const long long b = (long long)(k ^ 0x8000000000000000ULL);
  return __longlong_as_double(b >= 0 ? b : (b ^ 0x7fffffffffffffffLL));
}

// lane-per-target form (any automaton size): one lane evaluates the whole target serially
__global__ __launch_bounds__(kThreads) void k5_cyk_serial(LinArgs a) {
  __shared__ AutomatonLayout s_lay;
  stage_layout(a, &s_lay, kThreads);
  unsigned bx, by;
  swizzled_block(bx, by);
  LViews v(s_lay);
  make_lviews(a, by, v);
  __syncthreads();
  const int S = a.lay.S, d = a.d;
  if (d > v.q.W) return;
  const int t = bx * kThreads + threadIdx.x;
  if (t >= (v.q.L - d + 1) * S) return;
  const int i = t / S, s = t - i * S;
  TraceView R;
  R.ext = nullptr;
  const Constraint c{a.ys[v.n], a.ye[v.n], 1};
  cyk_target(v.m, v.q, v.in, R, c, d, i, s);
}

// staged form: same workgroup shape, context and operand staging as k4_in; KOWN (cell, tuple) products per lane.  Values only:
// the traceback re-derives the winner of the targets it visits (scan_rules.h, cyk_retrace), so the maxima of the two span-long
// candidate lists need no order bookkeeping here.
template <bool BIG, int KOWN>
__global__ __launch_bounds__(kBT, KOWN <= 2 ? ELEMDP_LB_CYK : 6) void k5_cyk(LinArgs a) {   // (KOWN 4: 83 registers, spills at 64)
  extern __shared__ double lds[];
  // (the automaton layout is read from the kernel arguments: constant offsets, scalar registers)

  unsigned bx, by;
  swizzled_block(bx, by);
  LViews v(a.lay);
  make_lviews(a, by, v);
  const AutomatonLayout& A = a.lay;
  const int S = a.lay.S, d = a.d, cpb = a.cpb, tid = threadIdx.x;
  if (d > v.q.W) return;
  const int ncell = v.q.L - d + 1, i0 = bx * cpb;
  if (i0 >= ncell) return;
  const int nc = (cpb < ncell - i0) ? cpb : ncell - i0;
  // lanes of the unary part and slots of the two maxima: the states that have a column in some plane (AutomatonLayout::st_live /
  // st_li) -- the others are log 0 in every plane, no list refers to them and nothing is stored for them
  const int NL = A.n_lane;
  const int CS = cpb * NL;
  const double NEG = ELEMDP_NEG_INF;
  unsigned long long* kb = reinterpret_cast<unsigned long long*>(lds);   // [CS] best key, rule 2
  unsigned long long* ke = kb + CS;                                      // [CS] best key, rule 6c
  double* st1 = lds + 2 * CS;                  // [KC][CU]  rows 1(i, i+a, front states)
  double* st2 = st1 + kChunkIn * kBT;     // [KC][CU]  rows 2(i+a, j, front states)
  const BlockLds BL = block_lds(2 * CS + 2 * kChunkIn * kBT, cpb, kLinEth + a.lay.n_theta, cpb + a.wmax + 3, staged_ints(a.lay, a.n_stage, 0));
  const BlockCtx cx = stage_context<BIG, 0>(a, v, reinterpret_cast<unsigned char*>(lds), BL, i0, nc, d, cpb);
  int* dm = cx.dm; int* cnts = cx.cnts; int* pre = cx.pre; int* base = cx.base;
  const int32_t* G = v.m.big;
  const unsigned long long kneg = cyk_key(NEG);
  for (int t = tid; t < CS; t += kBT) { kb[t] = kneg; ke[t] = kneg; }
  __syncthreads();
  // rule 2: candidates 1(i,i+a,s1) + 2(i+a,j,s2); a lane keeps the best of its tuples over the split points
  int a_lo = d;
  for (int c = 0; c < nc; ++c) { const int x = dm[c]; if (x > 0 && x < a_lo) a_lo = x; }
  const int nsp = A.n_split;
  const double* B = v.in.band;
  const StageMap sm = stage_map(A.n_front, S, cpb, nc, tid);
  const int NU = sm.NU, CU = sm.CU, KC = kChunkIn * sm.nsub;
  int po1[KOWN], po2[KOWN];
  double pv[KOWN];
#pragma unroll
  for (int r = 0; r < KOWN; ++r) {
    const int w = tid + r * kBT;
    po1[r] = -1; po2[r] = 0; pv[r] = NEG;
    if (w < nc * nsp) {
      const int c = w / nsp, t = w - c * nsp;
      const int x = dm[c];
      const int s1 = G[A.split_ent + 2 * t], s2 = G[A.split_ent + 2 * t + 1];
      // left_ok(i, d); a tuple with a state behind the front is log 0 in rule 2 (it serves rule 7)
      if (x > 0 && x <= d && s1 < NU && s2 < NU) { po1[r] = c * NU + s1; po2[r] = c * NU + s2; }
    }
  }
  const int kf = sm.act ? sm.r - sm.cell * NU : 0;      // the front state this lane stages
  const int fc1 = a.cyk_compact ? v.in.col(ST_1, kf) : 0, fc2 = a.cyk_compact ? v.in.col(ST_2, kf) : 0;
  for (int q0 = (a.dbg & 256) ? d : a_lo; q0 < d; q0 += KC) {   // (dbg 256 / 512: timing experiments without the split / item sums)
    if (sm.act) {
#pragma unroll
      for (int u = 0; u < kChunkIn; ++u) {
        const int q = q0 + u * sm.nsub + sm.sub;
        const bool ok = q < d;
        const int aa = ok ? q : q0;
        double x1, x2;
        if (a.cyk_compact) {      // (rows of the compact layout: the lane's front state through its columns in planes 1 and 2)
          x1 = B[fc1 >= 0 ? v.in.cidx(ST_1, aa, i0 + sm.cell, fc1) : 0u];
          x2 = B[fc2 >= 0 ? v.in.cidx(ST_2, d - aa, i0 + aa + sm.cell, fc2) : 0u];
          x1 = fc1 >= 0 ? x1 : NEG;
          x2 = fc2 >= 0 ? x2 : NEG;
        } else {
          x1 = B[v.in.idx(ST_1, aa, i0, 0) + sm.goff];
          x2 = B[v.in.idx(ST_2, d - aa, i0 + aa, 0) + sm.goff];
        }
        st1[(u * sm.nsub + sm.sub) * CU + sm.r] = ok ? x1 : NEG;
        st2[(u * sm.nsub + sm.sub) * CU + sm.r] = ok ? x2 : NEG;
      }
    }
    __syncthreads();
    // (staged slot v holds split point q0 + v, v = u * nsub + sub)
#pragma unroll
    for (int r = 0; r < KOWN; ++r)
      if (po1[r] >= 0) {
#pragma unroll 4
        for (int u = 0; u < KC; ++u) {
          const double y = st1[u * CU + po1[r]] + st2[u * CU + po2[r]];
          pv[r] = pv[r] < y ? y : pv[r];
        }
      }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < KOWN; ++r)
    if (po1[r] >= 0 && pv[r] != NEG) {
      const int w = tid + r * kBT;
      const int c = w / nsp, t = w - c * nsp;
      const int li = v.m.ints[A.st_li + G[A.split_tgt + t]];
      if (li >= 0) atomicMax(&kb[c * NL + li], cyk_key(pv[r]));
    }
  // rule 6c: candidates P(k,l,s1) + (L(i,k,s2) + (L(l,j,s3) + lam * tsc)) over the item records of the workgroup's cells (staged
  // in LDS) x the tuples
  const int nq = (a.dbg & 512) ? 0 : A.n_quad;
  {
    const int n_rec = outer_ranges(v, i0, nc, d, nq > 0, tid, cnts, pre, base);
    const OuterRecs R = outer_recs(st1, 2 * kChunkIn * kBT);
    {
      for (int p0 = 0; p0 < n_rec; p0 += R.cap) {
        const int np = (R.cap < n_rec - p0) ? R.cap : n_rec - p0;
        outer_stage<false>(v, R, p0, np, nc, tid, pre, base);
        const int total = np * nq;
        const int nqd = nq > 0 ? nq : 1, q256 = kBT / nqd, r256 = kBT % nqd;
        WorkIdx wi{tid / nqd, tid % nqd};
        for (int w0 = tid; w0 < total; w0 += kItemBatch * kBT) {
          double x0[kItemBatch], x1[kItemBatch], x2[kItemBatch], lt[kItemBatch];
          int hidx[kItemBatch];
          bool ok[kItemBatch];
#pragma unroll
          for (int u = 0; u < kItemBatch; ++u) {
            const int w = w0 + u * kBT;
            ok[u] = w < total;
            const int x = ok[u] ? wi.x : np - 1, t = ok[u] ? wi.t : nq - 1;
            wi.step(q256, r256, nqd);
            const LoopItem it = R.it[x];
            const int meta = R.meta[x];
            const int c = (meta >> 16) & 0x7fff;
            const int i = i0 + c, j = i + d;
            const int tgs = G[A.quad_tgt + t];
            ok[u] = ok[u] && meta >= 0;   // (sign bit: not in the inside set)
            x0[u] = v.in.ldm(ST_P, it.l - it.k, it.k, G[A.quad_ent + 3 * t]);
            x1[u] = v.in.ldm(ST_L, it.k - i, i, G[A.quad_ent + 3 * t + 1]);
            x2[u] = v.in.ldm(ST_L, j - it.l, it.l, G[A.quad_ent + 3 * t + 2]);
            lt[u] = ELEMDP_MUL_RN(v.m.lam(tgs), it.tsc);
            const int li = v.m.ints[A.st_li + tgs];
            ok[u] = ok[u] && li >= 0;
            hidx[u] = c * NL + (li >= 0 ? li : 0);
          }
#pragma unroll
          for (int u = 0; u < kItemBatch; ++u) {
            const double y = x0[u] + (x1[u] + (x2[u] + lt[u]));
            if (!ok[u] || y == NEG) continue;
            atomicMax(&ke[hidx[u]], cyk_key(y));
          }
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  if (tid < nc * NL) {
    const int c = tid / NL, s = v.m.ints[A.st_live + tid - c * NL];
    const int i = i0 + c;
    MaxAcc hB, hE;
    hB.best = cyk_unkey(kb[tid]);
    hE.best = cyk_unkey(ke[tid]);
    TraceView R;
    R.ext = nullptr;
    const Constraint con{a.ys[v.n], a.ye[v.n], 1};
    cyk_target_u(v.m, v.q, v.in, R, con, d, i, s, hB, hE);
  }
}

// exterior chain of the Viterbi pass and the traceback (one workgroup of 64 per sequence; lane 0 walks the trace).  Both are
// chains of dependent look-ups: with the sequence's context and the automaton blob in LDS (STAGE, as in k4_in_ext) only the
// table values themselves come from global memory.
template <bool STAGE>
__global__ __launch_bounds__(64) void k5_cyk_ext(LinArgs a) {
  extern __shared__ double l_cyk[];
  __shared__ AutomatonLayout s_lay;
  stage_layout(a, &s_lay, 64);
  LViews v(s_lay);
  make_lviews(a, blockIdx.x, v);
  stage_ext_context<STAGE>(a, v, reinterpret_cast<unsigned char*>(l_cyk), 0, 64);
  __syncthreads();
  const int S = a.lay.S, tid = threadIdx.x, L = v.q.L;
  TraceView R;
  R.ext = a.tr_ext + (size_t)blockIdx.x * a.ext_stride;
  const Constraint c{a.ys[v.n], a.ye[v.n], 1};
  for (int s = tid; s < S; s += 64) {
    v.in.o(0, s) = (s == a.lay.s00) ? 0. : ELEMDP_NEG_INF;
    TraceRec z; z.k = z.l = -1; z.t = -1; z.e1 = -1; z.s1 = -1;
    R.ext[s] = z;
  }
  __syncthreads();
  // a step: the lanes test the pair cells (i, j), i = j-1, j-2, .. (64 at a time); the few that are pairs with an exterior term
  // are then walked by the lanes as states, in the same order (descending i: the first strictly greatest candidate wins)
  const int W = v.q.W;
  for (int j = 1; j <= L; ++j) {
    for (int sb = 0; sb < S; sb += 64) {
      const int s = sb + tid;
      MaxAcc acc;
      for (int l0 = 0; l0 < W && l0 < j; l0 += 64) {
        const int dd = l0 + tid + 1, i = j - dd;
        double t = ELEMDP_NEG_INF;
        if (dd <= W && i >= 0 && v.q.pair_ok(i, dd)) t = v.q.e_ext[v.q.cell(i, dd)];
        unsigned long long kept = __ballot(t != ELEMDP_NEG_INF);
        while (kept) {
          const int l = __builtin_ctzll(kept);
          kept &= kept - 1;
          const double tl = __shfl(t, l, 64);
          if (s < S) cyk_ext_pair(v.m, v.in, acc, j, s, j - (l0 + l + 1), tl);
        }
      }
      if (s < S) cyk_ext_finish(v.m, v.q, v.in, R, c, j, s, acc);
    }
    __syncthreads();
  }
  int32_t* path = a.sc_psihat + v.seq_base;
  char* rss = a.sc_rss + v.seq_base;
  for (int p = tid; p < L; p += 64) { path[p] = 0; rss[p] = ' '; }
  __syncthreads();
  if (tid == 0) {
    const int s0 = v.in.o(L, a.lay.s0m2) < v.in.o(L, a.lay.s0m1) ? a.lay.s0m1 : a.lay.s0m2;
    TraceFrame* stack = reinterpret_cast<TraceFrame*>(a.trace_stack + (size_t)blockIdx.x * a.trace_stack_stride);
    trace_back(v.m, v.q, v.in, R, c, L, s0, path, rss, stack, (int)(a.trace_stack_stride * sizeof(int32_t) / sizeof(TraceFrame)));
  }
}

}  // namespace

hipError_t launch_lin_weights(const LinWeightArgs& a, hipStream_t st) {
  const size_t cells = a.cell_count ? a.cell_count : a.n_cells;
  const size_t work = (a.xwi && a.n_items > cells) ? a.n_items : cells;
  if (work == 0) return hipSuccess;
  size_t blocks = (work + kThreads - 1) / kThreads;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k4_weights, dim3((unsigned)blocks), dim3(kThreads), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_cyk_group(const LinArgs& full, int G, int Lmax, int Wmax, hipStream_t st) {
  if (G <= 0) return hipSuccess;
  LinArgs a = full;
  const int S = a.lay.S, nt = a.lay.n_theta;
  a.cpb = kBT / S;
  if (a.cpb > ELEMDP_CPB_MAX) a.cpb = ELEMDP_CPB_MAX;
  a.wmax = Wmax;
  a.ext_ring = G <= 1024 ? 1 : 0;
  a.lmax = Lmax;
  a.n_lin = kLinEth + nt; a.fast = 0; a.det = 0;
  a.cyk_compact = getenv("ELEMDP_CYK_DENSE") ? 0 : 1;   // (the dense table only for comparisons)
  // cells per workgroup: a lane per (cell, live state) in the unary part, the front rows of all cells in one staging row, at most
  // four (cell, split tuple) products per lane
  const int NLc = std::max(a.lay.n_lane, 1);
  a.cpb = std::min(std::min(kBT / NLc, kBT / std::max(a.lay.n_front, 1)), ELEMDP_CPB_MAX);
#ifndef ELEMDP_CYK_PROD
#define ELEMDP_CYK_PROD 4
#endif
  if (a.lay.n_split > 0) a.cpb = std::min(a.cpb, std::max(kBT / S, ELEMDP_CYK_PROD * kBT / a.lay.n_split));
  // (at least two split points per staging lane group: with the five front states of (.....) 32 cells left a third of the lanes idle
  // in every staging round and took twice the rounds -- 25 cells: scan 857 -> 821 ms per 10 000 x L=300)
  a.cpb = std::min(a.cpb, std::max(8, kBT / (2 * std::max(a.lay.n_front, 1))));
  if (const char* e = getenv("ELEMDP_CYK_CPB")) a.cpb = std::min(a.cpb, std::max(1, atoi(e)));   // (experiments)
  a.cpb = std::max(a.cpb, 1);
  const size_t lds = block_lds(2 * a.cpb * NLc + 2 * kChunkIn * kBT, a.cpb, kLinEth + nt, a.cpb + Wmax + 3, staged_ints(a.lay, a.n_stage, 0)).total;
  const bool big = a.n_stage >= a.lay.n_ints;
  const long long products = (long long)a.cpb * a.lay.n_split;   // (cell, tuple) products of a workgroup
  const int kown = products <= 2 * kBT ? 2 : products <= 4 * kBT ? 4 : products <= 8 * kBT ? 8 : 0;
  const bool staged = big && kown > 0 && !(a.dbg & 8) && lds <= 64 * 1024;
  for (int d = 0; d <= Wmax; ++d) {
    const int ncell = Lmax - d + 1;
    if (ncell <= 0) break;
    a.d = d;
    const dim3 grid((ncell + a.cpb - 1) / a.cpb, G);
    if (!staged) hipLaunchKernelGGL(k5_cyk_serial, dim3((ncell * S + kThreads - 1) / kThreads, G), dim3(kThreads), 0, st, a);
    else if (kown == 2) hipLaunchKernelGGL((k5_cyk<true, 2>), grid, dim3(kBT), lds, st, a);
    else if (kown == 4) hipLaunchKernelGGL((k5_cyk<true, 4>), grid, dim3(kBT), lds, st, a);
    else hipLaunchKernelGGL((k5_cyk<true, 8>), grid, dim3(kBT), lds, st, a);
  }
  const size_t lds_ext = (size_t)ext_lds(0, kLinEth + nt, Lmax, a.nword_max, a.n_stage).total;
  if (Lmax <= 2048 && a.nword_max <= 8192 && lds_ext <= 64 * 1024) hipLaunchKernelGGL(k5_cyk_ext<true>, dim3(G), dim3(64), lds_ext, st, a);
  else hipLaunchKernelGGL(k5_cyk_ext<false>, dim3(G), dim3(64), 0, st, a);
  return hipGetLastError();
}

// Blocks of cpb cells per band-kernel workgroup (LinArgs::nblk) for a diagonal of `nb` blocks: at most `nmax` (<= ELEMDP_CPB_MAX
// cells: the cell records and the stem mask of a workgroup), spread evenly over the fewest workgroups; one block per workgroup
// where the whole launch is resident at once anyway (small groups: there the lifetime of ONE workgroup is the launch's duration).
// DEFAULT: ONE block.  Three blocks per workgroup stage the context a third as often and measured the same on the bench
// (257.9 against 258.0 ms per step of 10 000 x L=200 on one box, 107.3 against 105.7 ms per 4096): the set-up they save is not
// what bounds the kernels, and they cost a resident workgroup per CU (LDS) -- DESIGN.md 4.2d.  Option "nblk" / ELEMDP_NBLK keep
// the form reachable (tests/test_round4_gpu.py runs it against the one-block kernels and the oracle).
static int env_nblk(int part) {   // experiments: ELEMDP_NBLK, or ELEMDP_NBLK_IN / ELEMDP_NBLK_OUT for one direction
  static const int v[3] = {[] { const char* e = getenv("ELEMDP_NBLK"); return e ? atoi(e) : 0; }(),
                           [] { const char* e = getenv("ELEMDP_NBLK_IN"); return e ? atoi(e) : 0; }(),
                           [] { const char* e = getenv("ELEMDP_NBLK_OUT"); return e ? atoi(e) : 0; }()};
  return v[1 + part] > 0 ? v[1 + part] : v[0];
}
#ifndef ELEMDP_NBLK_DEFAULT
#define ELEMDP_NBLK_DEFAULT 1
#endif
// req: LinArgs::nblk as the host engine passes it (option "nblk"): 0 = the policy above, n = n blocks wherever they fit;
// part 0: k4_in, 1: k4_out
static int nblk_max(int cpb, bool fast, bool det, int req, int part) {
  if (!fast || det) return 1;
  // default: up to ELEMDP_NBLK_DEFAULT blocks and ~40 cells per workgroup -- an automaton with few live states has large blocks
  // already (cpb = 32 for (.....): a second block there costs two of six resident workgroups and the scan got 15 % slower)
  const int want = req > 0 ? req : env_nblk(part) > 0 ? env_nblk(part) : std::min(ELEMDP_NBLK_DEFAULT, 40 / std::max(cpb, 1));
  return std::max(1, std::min(want, ELEMDP_CPB_MAX / std::max(cpb, 1)));
}
static void set_rcps(LinArgs& a, bool fast) {
  a.rcp_nap = 1.0f / (float)std::max(a.lay.n_ap, 1);
  a.rcp_lane = 1.0f / (float)std::max(fast ? a.lay.n_lane : a.lay.n_active, 1);
  a.rcp_cpb = 1.0f / (float)std::max(a.cpb, 1);
  a.rcp_3cpb = 1.0f / (float)std::max(3 * a.cpb, 1);
  // deterministic mode: lanes per cell in the pair phases = the power of two >= n_ap (no cell then straddles two waves)
  a.det_sh = -1;
  if (a.lay.n_ap <= 64) { a.det_sh = 0; while ((1 << a.det_sh) < a.lay.n_ap) ++a.det_sh; }
}
static int nblk_for(int nb, int G, int nmax, int req) {
  if (nmax <= 1 || (req <= 0 && (long long)nb * G <= 6144)) return 1;
  const int nsuper = (nb + nmax - 1) / nmax;
  return (nb + nsuper - 1) / nsuper;
}

hipError_t launch_lin_scan_group(const LinArgs& full, int G, int Lmax, int Wmax, int phase, hipStream_t st) {
  if (G <= 0) return hipSuccess;
  LinArgs a = full;
  const int S = a.lay.S, nt = a.lay.n_theta;
  a.cpb = kBT / S;
  if (a.cpb > ELEMDP_CPB_MAX) a.cpb = ELEMDP_CPB_MAX;
  a.wmax = Wmax;
  a.ext_ring = G <= 1024 ? 1 : 0;
  a.lmax = Lmax;
  a.schedule = 0;   // terminals (ari, nasi), Z = Z(ari,nasi): pass 0 of the reference schedule
  a.pass = 0;
  a.scan = 1;
  a.det = 0;
  const bool big = a.n_stage >= a.lay.n_ints;
  // table-driven unary phases as in launch_lin_group (the scanner's node tests are flag words of the fast blobs)
  const bool fast = a.fast && big && a.lay.fp_ok && !(a.dbg & 16) && a.lay.lin_total <= 2048;
  a.fast = fast ? 1 : 0;
  if (fast) a.cpb = std::min(kBT / std::max(a.lay.n_lane, 1), ELEMDP_CPB_MAX);
  a.n_lin = fast ? a.lay.lin_total : kLinEth + nt;
  const bool fp2 = a.lay.fp_max_p <= 2;
  const int hd = fast ? a.lay.n_lane : S;   // stride of the heavy sums per cell (k4_in / k4_out: HD)
  set_rcps(a, fast);
  // (the scan's sum passes keep one block per workgroup unless option "nblk" asks: with (.....) -- few live states, 18 cells per
  // block already -- a second block costs a resident workgroup per CU and measured 10 % slower, profiles/r04_*)
  const bool mb_scan = full.nblk > 0 || env_nblk(0) > 0 || env_nblk(1) > 0;
  const int nbmax_in = mb_scan ? nblk_max(a.cpb, fast, false, full.nblk, 0) : 1, nbmax_out = mb_scan ? nblk_max(a.cpb, fast, false, full.nblk, 1) : 1;
  auto lds_in_of = [&](int nblk) { const int ct = a.cpb * nblk; return (size_t)block_lds(2 * a.cpb * hd + kRecIn, ct, a.n_lin, ct + Wmax + 3, fast ? a.lay.fb_in_n : staged_ints(a.lay, a.n_stage, 0), 0, fast ? kCellInD : 0).total; };
  auto lds_out_of = [&](int nblk) { const int ct = a.cpb * nblk; return (size_t)block_lds(out_doubles(a.cpb * hd, nt, ct + Wmax + 3), ct, a.n_lin, ct + Wmax + 3, fast ? a.lay.fb_out_n : staged_ints(a.lay, a.n_stage, 1), 3 * ct, fast ? kCellOutD : 0).total; };
  const size_t lds_in = lds_in_of(1), lds_out = lds_out_of(1);
  // (the eighth k4_in workgroup of a CU fits the LDS: the 64-register variant; the sixth k4_out workgroup: the 80-register variant)
  auto w8_of = [&](int nblk) { return lds_in_of(nblk) * 8 <= 160 * 1024; };
  auto w6_of = [&](int nblk) { return lds_out_of(nblk) * 6 <= 160 * 1024; };
  if (getenv("ELEMDP_LDS_DEBUG") && phase == 0) fprintf(stderr, "scan group: G %d cpb %d S %d nt %d fast %d n_lin %d fast blob in/out %d/%d ints win %d lds k4_in %zu k4_out %zu; with %d / %d blocks per workgroup %zu / %zu\n", G, a.cpb, S, nt, (int)fast, a.n_lin, a.lay.fb_in_n, a.lay.fb_out_n, a.cpb + Wmax + 3, lds_in, lds_out, nbmax_in, nbmax_out, lds_in_of(nbmax_in), lds_out_of(nbmax_out));
  const bool stage_ext = Lmax <= 2048 && a.nword_max <= 8192;
  const size_t lds_ext_in = stage_ext ? (size_t)ext_lds((a.ext_ring ? ext_ring_doubles(0, Wmax, S, kLinEth + nt, Lmax, a.nword_max, a.n_stage) : 0), kLinEth + nt, Lmax, a.nword_max, a.n_stage).total : 0;
  const size_t lds_ext_out = stage_ext ? (size_t)ext_lds(2 * nt + 4 + (a.ext_ring ? ext_ring_doubles(2 * nt + 4, Wmax, S, kLinEth + nt, Lmax, a.nword_max, a.n_stage) : 0), kLinEth + nt, Lmax, a.nword_max, a.n_stage).total : sizeof(double) * (2 * nt + 4);
  const int ext_nt = (a.ext_block == kExtBlock && a.ext_ring && a.lay.n_active <= 128 && !(a.dbg & 8192)) ? 128 * kExtBlock : 128;   // (small groups: there the chain is exposed; in a large one its four-fold footprint only takes CUs from the band kernels)
#define ELEMDP_SCAN_PASS(CON, MODE)                                                                                              \
  do {                                                                                                                           \
    for (int d = 0; d <= Wmax; ++d) {                                                                                            \
      const int ncell = Lmax - d + 1;                                                                                            \
      if (ncell <= 0) break;                                                                                                     \
      a.d = d;                                                                                                                   \
      const int nb = (ncell + a.cpb - 1) / a.cpb;                                                                                \
      a.nblk = nblk_for(nb, G, nbmax_in, full.nblk);                                                                                           \
      const dim3 grid((nb + a.nblk - 1) / a.nblk, G);                                                                            \
      const size_t lds_i = lds_in_of(a.nblk);                                                                                    \
      if (a.nblk > 1 && fp2) hipLaunchKernelGGL((k4_in<true, CON, true, 2, false, true>), grid, dim3(kBT), lds_i, st, a);        \
      else if (a.nblk > 1) hipLaunchKernelGGL((k4_in<true, CON, true, kFastP, false, true>), grid, dim3(kBT), lds_i, st, a);     \
      else if (fast && fp2 && w8_of(a.nblk)) hipLaunchKernelGGL((k4_in<true, CON, true, 2, true>), grid, dim3(kBT), lds_i, st, a); \
      else if (fast && fp2) hipLaunchKernelGGL((k4_in<true, CON, true, 2>), grid, dim3(kBT), lds_i, st, a);                      \
      else if (fast) hipLaunchKernelGGL((k4_in<true, CON, true>), grid, dim3(kBT), lds_i, st, a);                                \
      else if (big) hipLaunchKernelGGL((k4_in<true, CON>), grid, dim3(kBT), lds_i, st, a);                                       \
      else hipLaunchKernelGGL((k4_in<false, CON>), grid, dim3(kBT), lds_i, st, a);                                               \
    }                                                                                                                            \
    if (stage_ext) hipLaunchKernelGGL((k4_in_ext<true, CON>), dim3(G), dim3(ext_nt), lds_ext_in, st, a);                         \
    else hipLaunchKernelGGL((k4_in_ext<false, CON>), dim3(G), dim3(128), 0, st, a);                                              \
    if (stage_ext) hipLaunchKernelGGL((k4_out_ext<MODE, true>), dim3(G), dim3(ext_nt), lds_ext_out, st, a);                      \
    else hipLaunchKernelGGL((k4_out_ext<MODE, false>), dim3(G), dim3(128), lds_ext_out, st, a);                                  \
    hipLaunchKernelGGL(k4_r7, dim3(((Lmax + 1) * (Wmax + 1) + kThreads - 1) / kThreads, G), dim3(kThreads), 0, st, a);           \
    for (int d = Wmax; d >= 0; --d) {                                                                                            \
      const int ncell = Lmax - d + 1;                                                                                            \
      if (ncell <= 0) continue;                                                                                                  \
      a.d = d;                                                                                                                   \
      const int nb = (ncell + a.cpb - 1) / a.cpb;                                                                                \
      a.nblk = nblk_for(nb, G, nbmax_out, full.nblk);                                                                                           \
      const dim3 grid((nb + a.nblk - 1) / a.nblk, G);                                                                            \
      const size_t lds_o = lds_out_of(a.nblk);                                                                                   \
      if (a.nblk > 1 && fp2) hipLaunchKernelGGL((k4_out<MODE, true, true, 2, false, true>), grid, dim3(kBT), lds_o, st, a);      \
      else if (a.nblk > 1) hipLaunchKernelGGL((k4_out<MODE, true, true, kFastP, false, true>), grid, dim3(kBT), lds_o, st, a);   \
      else if (fast && fp2 && w6_of(a.nblk)) hipLaunchKernelGGL((k4_out<MODE, true, true, 2, true>), grid, dim3(kBT), lds_o, st, a); \
      else if (fast && fp2) hipLaunchKernelGGL((k4_out<MODE, true, true, 2>), grid, dim3(kBT), lds_o, st, a);                    \
      else if (fast) hipLaunchKernelGGL((k4_out<MODE, true, true>), grid, dim3(kBT), lds_o, st, a);                              \
      else if (big) hipLaunchKernelGGL((k4_out<MODE, true>), grid, dim3(kBT), lds_o, st, a);                                     \
      else hipLaunchKernelGGL((k4_out<MODE, false>), grid, dim3(kBT), lds_o, st, a);                                             \
    }                                                                                                                            \
  } while (0)
  if (phase == 0) {
    ELEMDP_SCAN_PASS(false, OUT_SCAN);
    hipLaunchKernelGGL(k5_pick<0>, dim3((G + 63) / 64), dim3(64), 0, st, a, G);
  } else {
    ELEMDP_SCAN_PASS(true, OUT_END);
    hipLaunchKernelGGL(k5_pick<1>, dim3((G + 63) / 64), dim3(64), 0, st, a, G);
  }
#undef ELEMDP_SCAN_PASS
  return hipGetLastError();
}

hipError_t launch_lin_group(const LinArgs& full, int G, int Lmax, int Wmax, bool first_pass_only, hipStream_t st) {
  if (G <= 0) return hipSuccess;
  LinArgs a = full;
  const int S = a.lay.S, nt = a.lay.n_theta;
  a.cpb = kBT / S;
  if (a.cpb > ELEMDP_CPB_MAX) a.cpb = ELEMDP_CPB_MAX;
  a.wmax = Wmax;
  a.ext_ring = G <= 1024 ? 1 : 0;
  const bool big = a.n_stage >= a.lay.n_ints;
  // table-driven unary phases (lin_fast.h): the train schedule on an automaton whose lists fit the programs, the whole blob
  // and the weight tables staged
  const bool fast = a.fast && big && a.lay.fp_ok && !(a.dbg & 16) && a.lay.lin_total <= 2048;
  a.fast = fast ? 1 : 0;
  if (fast) a.cpb = std::min(kBT / std::max(a.lay.n_lane, 1), ELEMDP_CPB_MAX);   // (states without any column take no lane)
  a.n_lin = fast ? a.lay.lin_total : kLinEth + nt;
  const int NW = a.det ? kBT / 64 : 1;   // (copies of the statistics of k4_out: one per wave in the deterministic mode)
  const int hd = fast ? a.lay.n_lane : S;   // stride of the heavy sums per cell (k4_in / k4_out: HD)
  set_rcps(a, fast);
  const int nbmax_in = nblk_max(a.cpb, fast, a.det != 0, full.nblk, 0), nbmax_out = nblk_max(a.cpb, fast, a.det != 0, full.nblk, 1);
  auto lds_in_of = [&](int nblk) { const int ct = a.cpb * nblk; return (size_t)block_lds(2 * a.cpb * hd + kRecIn, ct, a.n_lin, ct + Wmax + 3, fast ? a.lay.fb_in_n : staged_ints(a.lay, a.n_stage, 0), 0, fast ? kCellInD : 0).total; };
  auto lds_out_of = [&](int nblk) { const int ct = a.cpb * nblk; return (size_t)block_lds(out_doubles(a.cpb * hd, nt, ct + Wmax + 3, NW, false), ct, a.n_lin, ct + Wmax + 3, fast ? a.lay.fb_out_n : staged_ints(a.lay, a.n_stage, 1), 3 * ct, fast ? kCellOutD : 0).total; };
  const size_t lds_in = lds_in_of(1);
  const size_t lds_stat = sizeof(double) * (2 * nt + 4);
  if (!a.no_rss)
    for (int d = 0; d <= Wmax; ++d) {
      const int ncell = Lmax - d + 1;
      if (ncell <= 0) break;
      a.d = d;
      const int nb = (ncell + a.cpb - 1) / a.cpb;
      a.nblk = nblk_for(nb, G, nbmax_in, full.nblk);
      const dim3 grid((nb + a.nblk - 1) / a.nblk, G);
      const size_t lds_i = lds_in_of(a.nblk);
      if (a.nblk > 1 && a.lay.fp_max_p <= 2) hipLaunchKernelGGL((k4_in<true, false, true, 2, false, true>), grid, dim3(kBT), lds_i, st, a);
      else if (a.nblk > 1) hipLaunchKernelGGL((k4_in<true, false, true, kFastP, false, true>), grid, dim3(kBT), lds_i, st, a);
      else if (fast && a.lay.fp_max_p <= 2 && lds_i * 8 <= 160 * 1024) hipLaunchKernelGGL((k4_in<true, false, true, 2, true>), grid, dim3(kBT), lds_i, st, a);
      else if (fast && a.lay.fp_max_p <= 2) hipLaunchKernelGGL((k4_in<true, false, true, 2>), grid, dim3(kBT), lds_i, st, a);
      else if (fast) hipLaunchKernelGGL((k4_in<true, false, true>), grid, dim3(kBT), lds_i, st, a);
      else if (big) hipLaunchKernelGGL((k4_in<true, false>), grid, dim3(kBT), lds_i, st, a);
      else hipLaunchKernelGGL((k4_in<false, false>), grid, dim3(kBT), lds_i, st, a);
    }
  a.nblk = 1;
  a.lmax = Lmax;
  const bool stage_ext = Lmax <= 2048 && a.nword_max <= 8192;
  const size_t lds_ext_in = stage_ext ? (size_t)ext_lds((a.ext_ring ? ext_ring_doubles(0, Wmax, S, kLinEth + nt, Lmax, a.nword_max, a.n_stage) : 0), kLinEth + nt, Lmax, a.nword_max, a.n_stage).total : 0;
  const int ext_nt = (a.ext_block == kExtBlock && a.ext_ring && a.lay.n_active <= 128 && !(a.dbg & 8192)) ? 128 * kExtBlock : 128;   // (small groups: there the chain is exposed; in a large one its four-fold footprint only takes CUs from the band kernels)
  if (stage_ext) hipLaunchKernelGGL((k4_in_ext<true, false>), dim3(G), dim3(ext_nt), lds_ext_in, st, a);
  else hipLaunchKernelGGL((k4_in_ext<false, false>), dim3(G), dim3(128), 0, st, a);
  // schedule 1 (automaton with the shadow state): ONE outside sweep carries both passes -- the "has motif" terminals on the
  // pattern's states, the "no motif" terminal on the shadow of (0,0), each with its own Z and statistics (lpass).
  // schedule 0: the reference's two sweeps, (ari, nasi) then the label's mask.
  const int n_pass = (a.schedule == 1 || first_pass_only) ? 1 : 2;
  const size_t lds_b = lds_out_of(1);
  if (getenv("ELEMDP_LDS_DEBUG")) fprintf(stderr, "lin group: G %d cpb %d fast %d n_lin %d staged ints in/out %d/%d lds k4_in %zu k4_out %zu; with %d / %d blocks per workgroup %zu / %zu\n", G, a.cpb, (int)fast, a.n_lin, staged_ints(a.lay, a.n_stage, 0), staged_ints(a.lay, a.n_stage, 1), lds_in, lds_b, nbmax_in, nbmax_out, lds_in_of(nbmax_in), lds_out_of(nbmax_out));
  for (int pass = 0; pass < n_pass; ++pass) {
    LinArgs b = a;
    b.pass = pass;
    const int n_stat = ext_stat_doubles(nt, b.det);
    if (stage_ext) hipLaunchKernelGGL((k4_out_ext<OUT_TRAIN, true>), dim3(G), dim3(b.det ? 128 : ext_nt), (size_t)ext_lds(n_stat + (b.ext_ring ? ext_ring_doubles(n_stat, Wmax, S, kLinEth + nt, Lmax, b.nword_max, b.n_stage) : 0), kLinEth + nt, Lmax, b.nword_max, b.n_stage).total, st, b);
    else hipLaunchKernelGGL((k4_out_ext<OUT_TRAIN, false>), dim3(G), dim3(128), sizeof(double) * n_stat, st, b);
    if (!b.no_rss) {
      hipLaunchKernelGGL(k4_r7, dim3(((Lmax + 1) * (Wmax + 1) + kThreads - 1) / kThreads, G), dim3(kThreads), 0, st, b);
      for (int d = Wmax; d >= 0; --d) {
        const int ncell = Lmax - d + 1;
        if (ncell <= 0) continue;
        b.d = d;
        const int nb = (ncell + b.cpb - 1) / b.cpb;
        b.nblk = nblk_for(nb, G, nbmax_out, full.nblk);
        const dim3 grid((nb + b.nblk - 1) / b.nblk, G);
        const size_t lds_o = lds_out_of(b.nblk);
        if (big && (b.dbg & 16)) hipLaunchKernelGGL((k4_out<OUT_NONE, true>), grid, dim3(kBT), lds_o, st, b);   // (timing experiment: no statistics)
        else if (b.nblk > 1 && b.lay.fp_max_p <= 2) hipLaunchKernelGGL((k4_out<OUT_TRAIN, true, true, 2, false, true>), grid, dim3(kBT), lds_o, st, b);
        else if (b.nblk > 1) hipLaunchKernelGGL((k4_out<OUT_TRAIN, true, true, kFastP, false, true>), grid, dim3(kBT), lds_o, st, b);
        else if (fast && b.lay.fp_max_p <= 2 && lds_o * 6 <= 160 * 1024) hipLaunchKernelGGL((k4_out<OUT_TRAIN, true, true, 2, true>), grid, dim3(kBT), lds_o, st, b);
        else if (fast && b.lay.fp_max_p <= 2) hipLaunchKernelGGL((k4_out<OUT_TRAIN, true, true, 2>), grid, dim3(kBT), lds_o, st, b);
        else if (fast) hipLaunchKernelGGL((k4_out<OUT_TRAIN, true, true>), grid, dim3(kBT), lds_o, st, b);
        else if (big) hipLaunchKernelGGL((k4_out<OUT_TRAIN, true>), grid, dim3(kBT), lds_o, st, b);
        else hipLaunchKernelGGL((k4_out<OUT_TRAIN, false>), grid, dim3(kBT), lds_o, st, b);
      }
    }
  }
  if ((a.schedule == 1 || a.lik_ratio || a.det) && !first_pass_only) hipLaunchKernelGGL(k4_combine, dim3(G), dim3(kThreads), 0, st, a, G);
  return hipGetLastError();
}

}  // namespace elemdp
