// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// Execution model: ONE WORKGROUP PER SEQUENCE (256 lanes = 4 waves, one per SIMD of a CU), the
// workgroups are persistent and pull sequences from a device-side queue (longest first).  A
// sequence's banded tables live in the workgroup's private HBM slot in the layout [e][d][i][s]
// (structural state, span, start, interval state): the lanes of a workgroup own the targets
// (i, s) of the current anti-diagonal d with s fastest, so every table access of a diagonal is a
// contiguous run of (L+1-d)*S doubles -- coalesced loads of the smaller diagonals, coalesced
// stores of the new one.  One barrier per diagonal; the exterior chain O(j) runs as S lanes.
// The automaton, theta, the pair mask, the sequence and its weights are staged in LDS; expected
// counts are accumulated with LDS fp64 atomics and leave the kernel once per sequence.
// No MFMA: this is a log-semiring recurrence, not a contraction (fp64 VALU + HBM bound).
//
// The arithmetic of every rule is in dp_rules.h / scan_rules.h / plan_rules.h (device code here).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "dp_rules.h"
#include "energy_rules.h"
#include "kernels.h"
#include "plan_rules.h"
#include "scan_rules.h"
#include "wave_gather.h"

namespace elemdp {

namespace {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_sum_int(int v, int* lds_tmp) {
  // wave reduction, then one LDS slot per wave
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds_tmp[wave] = v;
  __syncthreads();
  int tot = 0;
  for (int w = 0; w < kThreads / 64; ++w) tot += lds_tmp[w];
  __syncthreads();
  return tot;
}

// exclusive prefix sum of data[0..n) in place; data[n] receives the total.  One workgroup, tiles of one element per thread
// (coalesced loads whatever the array length, a wave scan per tile; round 3 gave every thread a contiguous chunk: 60 strided
// round trips per thread at 15 000 cells): `data` may be global or LDS memory, NT = threads of the workgroup (a multiple of 64),
// lds_tmp = NT / 64 + 1 ints.
template <int NT>
__device__ __forceinline__ void block_exclusive_scan_tiled(int32_t* data, int n, int* lds_tmp) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int run = 0;
  for (int base = 0; base < n; base += NT) {
    const int t = base + tid;
    const int v = (t < n) ? data[t] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) lds_tmp[wv] = inc;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { const int c = lds_tmp[w]; if (w < wv) before += c; tot += c; }
    if (t < n) data[t] = run + before + inc - v;
    run += tot;
    __syncthreads();
  }
  if (tid == 0) data[n] = run;
  __syncthreads();
}

struct OkBits {
  const uint32_t* bits; int L, W;
  __device__ __forceinline__ bool operator()(int i, int d) const {
    if (i < 0 || d < 0 || d > W || i + d > L) return false;
    const int c = i * (W + 1) + d;
    return (bits[c >> 5] >> (c & 31)) & 1u;
  }
};

// ---------------------------------------------------------------------------------------------
// K0: canonical pair mask (energy_model.hpp:213-219), one workgroup per sequence, one lane per word
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_mask(BatchArrays b, const SeqPlan* plans, int min_span, int write_bits,
                                                   uint32_t* okbits, int32_t* n_canonical) {
  __shared__ int tmp[kThreads / 64];
  const SeqPlan p = plans[blockIdx.x];
  const uint8_t* seq = b.seq + p.seq_base;
  const int ncell = (p.L + 1) * (p.W + 1);
  const int nword = (ncell + 31) / 32;
  int cnt = 0;
  for (int w = threadIdx.x; w < nword; w += kThreads) {
    uint32_t bits = 0;
    for (int k = 0; k < 32; ++k) {
      const int c = w * 32 + k;
      if (c >= ncell) break;
      const int i = c / (p.W + 1), d = c - i * (p.W + 1);
      if (canonical_pair(seq, p.L, p.W, min_span, i, d)) { bits |= 1u << k; ++cnt; }
    }
    if (write_bits) okbits[p.bits_base + w] = bits;
  }
  const int tot = block_sum_int(cnt, tmp);
  if (threadIdx.x == 0) n_canonical[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------------
// Plan builder.  All kernels: grid (chunks of 256 cells / words / items, sequences of the plan set), so that a set of a
// few hundred long sequences still fills the GPU; per-sequence prefix sums are one workgroup per sequence.
// ---------------------------------------------------------------------------------------------
// the pair mask indexed by (end l, span d): bit l * (W+1) + d <=> pair cell (l - d, d)   (see enum_interior_by_end)
__global__ __launch_bounds__(kThreads) void k_mask_by_end(PlanKernelArgs a) {
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int L = p.L, W = p.W;
  const int ncell = (L + 1) * (W + 1), nword = (ncell + 31) / 32;
  const int w = blockIdx.x * kThreads + threadIdx.x;
  if (w >= nword) return;
  const OkBits ok{a.okbits + p.bits_base, L, W};
  uint32_t bits = 0;
  for (int b = 0; b < 32; ++b) {
    const int c = w * 32 + b;
    if (c >= ncell) break;
    const int l = c / (W + 1), d = c - l * (W + 1);
    if (l - d >= 0 && ok(l - d, d)) bits |= 1u << b;
  }
  a.okbits_end[p.bits_base + w] = bits;
}

struct EndWords {
  const uint32_t* bits; int nword;
  __device__ __forceinline__ uint32_t operator()(int n) const { return n < nword ? bits[n] : 0u; }
};

// The E cells that have interior loops at all (their closing pair is kept) are ~8 % of the cells of a filtered mask: a lane per
// cell leaves a wave with a handful of busy lanes.  A workgroup therefore takes kPlanTile * kThreads consecutive cells, compacts
// the ones with loops into an LDS list (ballot + popcount per wave) and hands THOSE to its first lanes.
constexpr int kPlanTile = 4;
struct LoopCells {
  int list[kPlanTile * kThreads];
  int wcnt[kPlanTile][kThreads / 64];
  int n;
};
// has_loops(c): cell c of the sequence is an E cell whose closing pair (i-1, d+2) is kept.  Returns the number of such cells of the
// workgroup's tile, their cell indices in sh.list (ascending).
template <class Pred> __device__ __forceinline__ int compact_loop_cells(LoopCells& sh, int ncell, Pred has_loops) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  bool act[kPlanTile];
  unsigned long long m[kPlanTile];
#pragma unroll
  for (int r = 0; r < kPlanTile; ++r) {
    const int c = (blockIdx.x * kPlanTile + r) * kThreads + tid;
    act[r] = c < ncell && has_loops(c);
    m[r] = __ballot(act[r]);
    if (lane == 0) sh.wcnt[r][wv] = __popcll(m[r]);
  }
  __syncthreads();
  int before = 0;
#pragma unroll
  for (int r = 0; r < kPlanTile; ++r) {
    for (int w = 0; w < kThreads / 64; ++w) {
      const int n = sh.wcnt[r][w];
      if (w == wv && act[r]) sh.list[before + __popcll(m[r] & ((1ull << lane) - 1ull))] = (blockIdx.x * kPlanTile + r) * kThreads + tid;
      before += n;
    }
  }
  __syncthreads();
  return before;
}

// K-plan 1: dmin, structural terms of every kept pair, interior-loop counts per E cell
__global__ __launch_bounds__(kThreads) void k_plan_cells(PlanKernelArgs a, int32_t* n_items_out) {
  __shared__ int tmp[kThreads / 64];
  __shared__ LoopCells sh;
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int L = p.L, W = p.W;
  const int ncell = (L + 1) * (W + 1);
  if ((int)(blockIdx.x * kPlanTile * kThreads) >= ncell) return;
  const uint8_t* seq = a.b.seq + p.seq_base;
  const int32_t* ndot = a.fix_rss ? a.b.ndot + p.pos_base : nullptr;
  const OkBits ok{a.okbits + p.bits_base, L, W};
  const EndWords words{a.okbits_end + p.bits_base, (ncell + 31) / 32};
  const PlanCfg cfg{a.no_ene, a.min_span, a.fix_rss};
  if (blockIdx.x == 0) {
    int16_t* dmin = a.p.dmin + p.dmin_base;
    for (int i = threadIdx.x; i <= L; i += kThreads) {
      int dm = 0;
      for (int d = 1; d <= W && i + d <= L; ++d)
        if (ok(i, d)) { dm = d; break; }
      dmin[i] = (int16_t)dm;
    }
  }
  int32_t* cnt = a.p.by_outer_off + p.off_base;
  auto has_loops = [&](int c) {
    const int i = c / (W + 1), d = c - i * (W + 1);
    return i + d <= L && i > 0 && d + 2 <= W && ok(i - 1, d + 2);
  };
#pragma unroll 1
  for (int r = 0; r < kPlanTile; ++r) {
    const int c = (blockIdx.x * kPlanTile + r) * kThreads + threadIdx.x;
    if (c >= ncell) continue;
    const int i = c / (W + 1), d = c - i * (W + 1);
    if (ok(i, d)) {
      const PairTerms t = pair_terms(*a.et, cfg, seq, L, ndot, i, d, ok(i + 1, d - 2));
      const size_t g = p.cell_base + c;
      a.p.e_stack[g] = t.stack; a.p.e_ext[g] = t.ext; a.p.e_ml[g] = t.ml; a.p.e_close[g] = t.close; a.p.e_hp[g] = t.hp;
    }
    if (!has_loops(c)) cnt[c] = 0;
  }
  const int n_act = compact_loop_cells(sh, ncell, has_loops);
  bool fast = a.count_fast != 0;      // (an N next to a pair makes a 2x2 loop log 0: such sequences are enumerated)
  if (fast) {
    int has_n = 0;
    for (int t = threadIdx.x; t < L; t += kThreads) has_n |= (seq[t] == 0);
    fast = !__syncthreads_or(has_n);
  }
  int n_mine = 0;
  for (int t = threadIdx.x; t < n_act; t += kThreads) {
    const int c = sh.list[t];
    const int i = c / (W + 1), d = c - i * (W + 1);
    int n = 0;
    if (fast) n = count_interior_by_end(a.no_ene != 0, L, W, p.C, words, i, d);
    else enum_interior_by_end(*a.et, cfg, seq, L, W, p.C, ndot, words, i, d, [&](int, int, double, bool) { ++n; });
    cnt[c] = n;
    n_mine += n;
  }
  const int tot = block_sum_int(n_mine, tmp);
  if (threadIdx.x == 0 && tot) atomicAdd(&n_items_out[blockIdx.y], tot);
}

// K-plan 2: item lists.  CSR by outer cell (deterministic enumeration: the reference's order), then three index lists
// (by inner pair, by left loop, by right loop) via counting sort; segments are sorted by item index so that the
// summation order of the gathers is reproducible.
__device__ __forceinline__ int role_key(const LoopItem& it, int role, int W) {
  switch (role) {
    case 0: return it.k * (W + 1) + (it.l - it.k);  // inner pair cell (k, l)
    case 1: return it.i * (W + 1) + (it.k - it.i);  // left loop cell (i, k)
    default: return it.l * (W + 1) + (it.j - it.l); // right loop cell (l, j)
  }
}
__device__ __forceinline__ int32_t* role_off(const PlanArrays& p, int role) {
  return role < 0 ? p.by_outer_off : role == 0 ? p.by_inner_off : role == 1 ? p.by_left_off : p.by_right_off;
}
__device__ __forceinline__ int32_t* role_idx(const PlanArrays& p, int role) {
  return role == 0 ? p.by_inner_idx : role == 1 ? p.by_left_idx : p.by_right_idx;
}

// counts -> offsets of one CSR array per sequence (role -1: by_outer)
__global__ __launch_bounds__(kThreads) void k_plan_scan(PlanKernelArgs a, int role) {
  __shared__ int tmp[kThreads / 64 + 1];
  const SeqPlan p = a.plans[a.first + blockIdx.x];
  block_exclusive_scan_tiled<kThreads>(role_off(a.p, role) + p.off_base, (p.L + 1) * (p.W + 1), tmp);
}

__global__ __launch_bounds__(kThreads) void k_plan_fill(PlanKernelArgs a) {
  __shared__ LoopCells sh;
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int L = p.L, W = p.W;
  const int ncell = (L + 1) * (W + 1);
  if ((int)(blockIdx.x * kPlanTile * kThreads) >= ncell) return;
  const OkBits ok{a.okbits + p.bits_base, L, W};
  const int n_act = compact_loop_cells(sh, ncell, [&](int c) {
    const int i = c / (W + 1), d = c - i * (W + 1);
    return i + d <= L && i > 0 && d + 2 <= W && ok(i - 1, d + 2);
  });
  const uint8_t* seq = a.b.seq + p.seq_base;
  const int32_t* ndot = a.fix_rss ? a.b.ndot + p.pos_base : nullptr;
  const EndWords words{a.okbits_end + p.bits_base, (ncell + 31) / 32};
  const PlanCfg cfg{a.no_ene, a.min_span, a.fix_rss};
  LoopItem* items = a.p.items + p.item_base;
  uint8_t* item_in = a.p.item_in + p.item_base;
  for (int t = threadIdx.x; t < n_act; t += kThreads) {
    const int c = sh.list[t];
    const int i = c / (W + 1), d = c - i * (W + 1);
    int pos = a.p.by_outer_off[p.off_base + c];
    enum_interior_by_end(*a.et, cfg, seq, L, W, p.C, ndot, words, i, d, [&](int k, int l, double tsc, bool in) {
      LoopItem it;
      it.tsc = tsc; it.i = (int16_t)i; it.j = (int16_t)(i + d); it.k = (int16_t)k; it.l = (int16_t)l;
      items[pos] = it;
      item_in[pos] = in ? 1 : 0;
      ++pos;
    });
  }
}

// one role: zero the counters | count the keys | (k_plan_scan) | scatter the item indices | sort every segment
__global__ __launch_bounds__(kThreads) void k_role_zero(PlanKernelArgs a, int role) {
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c > (p.L + 1) * (p.W + 1)) return;
  role_off(a.p, role)[p.off_base + c] = 0;
  a.p.cursor[p.off_base + c] = 0;
}
__global__ __launch_bounds__(kThreads) void k_role_count(PlanKernelArgs a, int role) {
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= p.n_items) return;
  atomicAdd(&role_off(a.p, role)[p.off_base + role_key(a.p.items[p.item_base + t], role, p.W)], 1);
}
__global__ __launch_bounds__(kThreads) void k_role_scatter(PlanKernelArgs a, int role) {
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= p.n_items) return;
  const int key = role_key(a.p.items[p.item_base + t], role, p.W);
  const int pos = atomicAdd(&a.p.cursor[p.off_base + key], 1);
  role_idx(a.p, role)[p.item_base + role_off(a.p, role)[p.off_base + key] + pos] = t;
}
// The three roles in one pass each (plans that keep item copies in the secondary orders: PlanArrays::items_inner .. are allocated
// before the lists are built and serve as scratch here).  The count pass keeps the RANK its atomicAdd returns -- the position of
// the item among the items of its key -- so the scatter pass needs no cursor and no atomics, and an item is read once per pass
// instead of once per role and pass: 12 launches -> 6, six atomics per item -> three.
__global__ __launch_bounds__(kThreads) void k_role_zero3(PlanKernelArgs a) {
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c > (p.L + 1) * (p.W + 1)) return;
  a.p.by_inner_off[p.off_base + c] = 0;
  a.p.by_left_off[p.off_base + c] = 0;
  a.p.by_right_off[p.off_base + c] = 0;
}
__global__ __launch_bounds__(kThreads) void k_role_count3(PlanKernelArgs a) {
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= p.n_items) return;
  const LoopItem it = a.p.items[p.item_base + t];
  int32_t* rank = reinterpret_cast<int32_t*>(a.p.items_inner) + (size_t)(p.item_base + t) * 3;
#pragma unroll
  for (int role = 0; role < 3; ++role) rank[role] = atomicAdd(&role_off(a.p, role)[p.off_base + role_key(it, role, p.W)], 1);
}
__global__ __launch_bounds__(kThreads) void k_role_scatter3(PlanKernelArgs a) {
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= p.n_items) return;
  const LoopItem it = a.p.items[p.item_base + t];
  const int32_t* rank = reinterpret_cast<const int32_t*>(a.p.items_inner) + (size_t)(p.item_base + t) * 3;
#pragma unroll
  for (int role = 0; role < 3; ++role)
    role_idx(a.p, role)[p.item_base + role_off(a.p, role)[p.off_base + role_key(it, role, p.W)] + rank[role]] = t;
}
// One role of one sequence in ONE workgroup of 1024 threads (round 4): the counters of the keys live in LDS (4 bytes per cell: a
// sequence of 300 with a band of 50 takes 61 KB), so the count pass is a pass of LDS atomics -- the global atomics of k_role_count3
// serialise on the loop cells that hundreds of items share (32 ms per 10 000 x L=300) -- the scan runs on the LDS array, and the
// scatter takes its positions from a second pass of LDS atomics on the scanned array (offsets written out before it).  COPY: the
// scatter also writes the item's copy in the role's order (k_permute_items: a scattered 16-byte read per item and role otherwise).
// grid = (3 roles, sequences); dynamic LDS = (cells + 1) ints; the launcher falls back to the passes above for sequences whose
// counters do not fit.
constexpr int kRoleThreads = 1024;
template <bool COPY>
__global__ __launch_bounds__(kRoleThreads) void k_role_build(PlanKernelArgs a) {
  extern __shared__ int32_t s_cnt[];
  __shared__ int tmp[kRoleThreads / 64 + 1];
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int role = blockIdx.x;
  const int ncell = (p.L + 1) * (p.W + 1), n_items = p.n_items, W = p.W;
  const int tid = threadIdx.x;
  for (int c = tid; c <= ncell; c += kRoleThreads) s_cnt[c] = 0;
  __syncthreads();
  const LoopItem* items = a.p.items + p.item_base;
  for (int t = tid; t < n_items; t += kRoleThreads) atomicAdd(&s_cnt[role_key(items[t], role, W)], 1);
  __syncthreads();
  block_exclusive_scan_tiled<kRoleThreads>(s_cnt, ncell, tmp);
  int32_t* off = role_off(a.p, role) + p.off_base;
  for (int c = tid; c <= ncell; c += kRoleThreads) off[c] = s_cnt[c];
  __syncthreads();
  int32_t* idx = role_idx(a.p, role) + p.item_base;
  LoopItem* copy = (role == 0 ? a.p.items_inner : role == 1 ? a.p.items_left : a.p.items_right) + p.item_base;
  for (int t = tid; t < n_items; t += kRoleThreads) {
    const LoopItem it = items[t];
    const int pos = atomicAdd(&s_cnt[role_key(it, role, W)], 1);
    idx[pos] = t;
    if (COPY) copy[pos] = it;
  }
}
// Sorts every segment of one role by item index (the scatter above leaves them in arbitrary order).  The segments of the
// cells (i, 0..W) of one row are contiguous in the CSR array: a workgroup takes a row, builds the composite values
// (cell << 32 | item index) in LDS, bitonic-sorts the whole row -- the cells are already grouped, so this sorts inside every
// segment, and the hundreds of items that share an empty loop cell (i, 0) are sorted by all lanes instead of one --
// and writes the indices back.  CAP = LDS capacity in elements; a launch handles the rows with CAP_LO < n <= CAP, longer
// rows (only with the BPP filter off on long sequences) fall back to a heapsort per segment in global memory.
template <int CAP, int CAP_LO, class KEY>
__global__ __launch_bounds__(kThreads) void k_role_sort(PlanKernelArgs a, int role) {
  // KEY = uint32_t: cell in the top 6 bits, item index below (W + 1 <= 64 and fewer than 2^26 items per sequence: half the
  // LDS, twice the workgroups per CU); uint64_t otherwise
  constexpr int kShift = sizeof(KEY) == 4 ? 26 : 32;
  __shared__ KEY sv[CAP];
  __shared__ int srow[1024 + 2];
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const int i = blockIdx.x, W1 = p.W + 1;
  if (i > p.L) return;
  const int32_t* roff = role_off(a.p, role) + p.off_base + (size_t)i * W1;
  int32_t* ridx = role_idx(a.p, role) + p.item_base;
  const int row0 = roff[0], n = roff[W1] - row0;
  if (n <= CAP_LO || n <= 1) return;
  const int tid = threadIdx.x;
  if (n > CAP || W1 > 1024) {
    if (CAP < 8192) return;   // (the launch with the largest capacity takes the oversized rows)
    for (int c = tid; c < W1; c += kThreads) {
      int32_t* v = ridx + roff[c];
      const int m = roff[c + 1] - roff[c];
      auto sift = [&](int root, int end) {
        const int e = v[root];
        for (;;) {
          int ch = 2 * root + 1;
          if (ch >= end) break;
          if (ch + 1 < end && v[ch + 1] > v[ch]) ++ch;
          if (v[ch] <= e) break;
          v[root] = v[ch];
          root = ch;
        }
        v[root] = e;
      };
      for (int r = m / 2 - 1; r >= 0; --r) sift(r, m);
      for (int end = m - 1; end > 0; --end) {
        const int t = v[0]; v[0] = v[end]; v[end] = t;
        sift(0, end);
      }
    }
    return;
  }
  for (int c = tid; c <= W1; c += kThreads) srow[c] = roff[c] - row0;
  int np = 2;
  while (np < n) np <<= 1;
  __syncthreads();
  for (int x = tid; x < np; x += kThreads) {
    KEY e = (KEY)~(KEY)0;
    if (x < n) {
      int lo = 0, hi = W1 - 1;   // the cell whose segment holds position x: largest c with srow[c] <= x
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (srow[mid] <= x) lo = mid; else hi = mid - 1;
      }
      e = ((KEY)lo << kShift) | (KEY)(unsigned)ridx[row0 + x];
    }
    sv[x] = e;
  }
  __syncthreads();
  for (int k = 2; k <= np; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int x = tid; x < np; x += kThreads) {
        const int y = x ^ j;
        if (y > x) {
          const KEY u = sv[x], w = sv[y];
          if ((u > w) == ((x & k) == 0)) { sv[x] = w; sv[y] = u; }
        }
      }
      __syncthreads();
    }
  for (int x = tid; x < n; x += kThreads) ridx[row0 + x] = (int32_t)(sv[x] & (KEY)((((KEY)1) << kShift) - 1));
}

// ---------------------------------------------------------------------------------------------
// one sequence worth of sweeps (all lanes of the workgroup call these together)
// ---------------------------------------------------------------------------------------------
struct Prof {
  long long* p; long long t;
  __device__ __forceinline__ void start() { if (p && threadIdx.x == 0) t = clock64(); }
  __device__ __forceinline__ void lap(int k) { if (p && threadIdx.x == 0) { long long n = clock64(); p[k] += n - t; t = n; } }
};

// per-sequence scratch shared by the sweeps
struct SweepScratch {
  int* cell_counter;   // LDS: next cell of the heavy phase
  double* wave_scr;    // LDS: kWaves * 128 doubles
  double* tmp;         // global: 3 * tmp_stride doubles per slot
  size_t tmp_stride;
};

template <bool CONSTRAINED>
__device__ void sweep_inside(const ModelView& m, const SeqView& q, const TableView& T, const Constraint& c, bool no_rss,
                             const SweepScratch& sc, Prof& pf) {
  const int S = m.lay.S, tid = threadIdx.x;
  WaveCtx w;
  w.lane = tid & 63;
  w.scr_m = sc.wave_scr + (tid >> 6) * 128;
  w.scr_s = w.scr_m + 64;
  if (!no_rss) {
    if (tid == 0) *sc.cell_counter = 0;
    __syncthreads();
    for (int d = 0; d <= q.W; ++d) {
      // heavy phase: waves pull cells
      const int ncell = q.L - d + 1;
      for (;;) {
        int i = 0;
        if (w.lane == 0) i = atomicAdd(sc.cell_counter, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= ncell) break;
        heavy_inside_cell(m, q, T, w, d, i, sc.tmp);
      }
      __syncthreads();
      pf.lap(6);
      if (tid == 0) *sc.cell_counter = 0;
      const int n = ncell * S;
      for (int t = tid; t < n; t += kThreads) {
        const int i = t / S, s = t - i * S;
        const double HB = q.left_ok(i, d) ? T.at(ST_B, d, i, s) : ELEMDP_NEG_INF;
        const double HE = q.e_ok(i, d) ? sc.tmp[(size_t)i * S + s] : ELEMDP_NEG_INF;
        inside_target_u<CONSTRAINED>(m, q, T, c, d, i, s, HB, HE);
      }
      __syncthreads();
      pf.lap(1);
    }
  }
  for (int s = tid; s < S; s += kThreads) T.o(0, s) = (s == m.lay.s00) ? 0. : ELEMDP_NEG_INF;
  __syncthreads();
  for (int j = 1; j <= q.L; ++j) {
    for (int s = tid; s < S; s += kThreads) inside_ext_target<CONSTRAINED>(m, q, T, c, j, s);
    __syncthreads();
  }
  pf.lap(2);
}

template <int MODE>
__device__ void sweep_outside(const ModelView& m, const SeqView& q, const TableView& in, const TableView& out, double Z,
                              const Constraint& c, bool ari, bool nasi, GpuSink& sink, double* lds_eh, bool no_rss,
                              const SweepScratch& sc, Prof& pf) {
  const int S = m.lay.S, tid = threadIdx.x;
  OutCtx<GpuSink> x{m, q, in, out, Z, c, sink};
  sink.eh0 = sink.eh1 = 0.;
  WaveCtx w;
  w.lane = tid & 63;
  w.scr_m = sc.wave_scr + (tid >> 6) * 128;
  w.scr_s = w.scr_m + 64;
  for (int s = tid; s < S; s += kThreads) {
    double v = ELEMDP_NEG_INF;
    if (nasi && s == m.lay.s00) v = 0.;
    if (ari && (s == m.lay.s0m1 || s == m.lay.s0m2)) v = 0.;
    out.o(q.L, s) = v;
  }
  if (tid == 0) *sc.cell_counter = 0;
  __syncthreads();
  for (int i = q.L - 1; i >= 0; --i) {
    for (int s = tid; s < S; s += kThreads) outside_ext_target<MODE>(x, i, s);
    __syncthreads();
  }
  pf.lap(3);
  if (!no_rss) {
    for (int d = q.W; d >= 0; --d) {
      const int ncell = q.L - d + 1;
      for (;;) {
        int i = 0;
        if (w.lane == 0) i = atomicAdd(sc.cell_counter, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= ncell) break;
        heavy_outside_cell<MODE>(x, w, d, i, sc.tmp, sc.tmp_stride);
      }
      __syncthreads();
      pf.lap(7);
      if (tid == 0) *sc.cell_counter = 0;
      const int n = ncell * S;
      for (int t = tid; t < n; t += kThreads) {
        const int i = t / S, s = t - i * S;
        HeavyOut H;
        const bool lok = q.left_ok(i, d);
        H.H1 = lok ? out.at(ST_1, d, i, s) : ELEMDP_NEG_INF;
        H.H2 = lok ? sc.tmp[0 * sc.tmp_stride + (size_t)i * S + s] : ELEMDP_NEG_INF;
        H.HP = q.pair_ok(i, d) ? sc.tmp[1 * sc.tmp_stride + (size_t)i * S + s] : ELEMDP_NEG_INF;
        H.HL = sc.tmp[2 * sc.tmp_stride + (size_t)i * S + s];
        outside_target_u<MODE>(x, d, i, s, H);
      }
      __syncthreads();
      pf.lap(4);
    }
  }
  if (MODE == OUT_TRAIN) {
    const double a = wave_sum(sink.eh0), b = wave_sum(sink.eh1);
    if ((tid & 63) == 0) { atomicAdd(&lds_eh[0], a); atomicAdd(&lds_eh[1], b); }
    __syncthreads();
  }
}

__device__ void sweep_cyk(const ModelView& m, const SeqView& q, const TableView& T, const TraceView& R,
                                       const Constraint& c) {
  const int S = m.lay.S, tid = threadIdx.x;
  for (int d = 0; d <= q.W; ++d) {
    const int nn = (q.L - d + 1) * S;
    for (int t = tid; t < nn; t += kThreads) { const int i = t / S, s = t - i * S; cyk_target(m, q, T, R, c, d, i, s); }
    __syncthreads();
  }
  for (int s = tid; s < S; s += kThreads) {
    T.o(0, s) = (s == m.lay.s00) ? 0. : ELEMDP_NEG_INF;
    TraceRec leaf; leaf.k = leaf.l = -1; leaf.t = -1; leaf.e1 = -1; leaf.s1 = -1;
    R.ext[s] = leaf;
  }
  __syncthreads();
  for (int j = 1; j <= q.L; ++j) {
    for (int s = tid; s < S; s += kThreads) cyk_ext_target(m, q, T, R, c, j, s);
    __syncthreads();
  }
}

__device__ bool run_trace_back(const ModelView& m, const SeqView& q, const TableView& T, const TraceView& R, const Constraint& c,
                               int L, int s0, int32_t* path, char* rss, TraceFrame* stack, int cap) {
  return trace_back(m, q, T, R, c, L, s0, path, rss, stack, cap);
}

// ---------------------------------------------------------------------------------------------
// the DP kernel: TRAIN = K2 + 2 x K3 fused, BPP = K1, SCAN = K4 + K5 + K6
// ---------------------------------------------------------------------------------------------
#ifndef ELEMDP_MIN_WAVES
#define ELEMDP_MIN_WAVES 2
#endif
template <int KIND>
__global__ __launch_bounds__(kThreads, ELEMDP_MIN_WAVES) void k_dp(DpArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];
  __shared__ int l_cur;
  int32_t* l_ints = reinterpret_cast<int32_t*>(lds + a.lds.ints);
  double* l_theta = reinterpret_cast<double*>(lds + a.lds.theta);
  double* l_en_o = reinterpret_cast<double*>(lds + a.lds.en_o);
  double* l_en_x = reinterpret_cast<double*>(lds + a.lds.en_x);
  double* l_eh = reinterpret_cast<double*>(lds + a.lds.eh);   // EHo[2], EHx[2]
  double* l_zs = reinterpret_cast<double*>(lds + a.lds.zs);   // Zo, Zari, Znasi, skip, Ys, Ye
  double* l_ws = reinterpret_cast<double*>(lds + a.lds.ws);
  double* l_post = reinterpret_cast<double*>(lds + a.lds.post);
  uint32_t* l_ok = reinterpret_cast<uint32_t*>(lds + a.lds.okbits);
  int16_t* l_dmin = reinterpret_cast<int16_t*>(lds + a.lds.dmin);
  uint8_t* l_seq = lds + a.lds.seq;
  uint8_t* l_unp = lds + a.lds.unp;
  const int tid = threadIdx.x;
  const int S = a.lay.S, nt = a.lay.n_theta;

  for (int t = tid; t < a.lay.n_small; t += kThreads) l_ints[t] = a.ints[t];
  const ParamBlock* pb = reinterpret_cast<const ParamBlock*>(a.params);
  const double* g_theta = a.params + sizeof(ParamBlock) / sizeof(double);
  for (int t = tid; t < nt; t += kThreads) l_theta[t] = g_theta[t];

  ModelView m(*a.layp);
  m.ints = l_ints;
  m.big = a.ints;
  m.theta = l_theta;
  m.lambda[0] = pb->lambda[0];
  m.lambda[1] = pb->lambda[1];
  m.log_tau = pb->log_tau;
  m.lam_same = pb->lam_same;
  m.no_prf = a.no_prf;
  m.m_min = a.m_min;
  const bool no_rss = a.no_rss != 0;
  __shared__ int l_cell_counter;
  SweepScratch sc;
  sc.cell_counter = &l_cell_counter;
  sc.wave_scr = reinterpret_cast<double*>(lds + a.lds.wave_scr);
  sc.tmp_stride = a.tmp_stride;
  sc.tmp = a.tmp + (size_t)blockIdx.x * 3 * a.tmp_stride;
  Prof pf;
  pf.p = a.prof ? a.prof + (size_t)blockIdx.x * 8 : nullptr;
  pf.t = 0;
  pf.start();

  for (;;) {
    pf.lap(5);
    __syncthreads();
    if (tid == 0) l_cur = atomicAdd(a.counter, 1);
    __syncthreads();
    const int w = l_cur;
    if (w >= a.n_seq) break;
    const int n = a.order[w];
    const SeqPlan p = a.plans[n];
    const int L = p.L, W = p.W;
    const int ncell = (L + 1) * (W + 1);
    const int nword = (ncell + 31) / 32;

    // ---- stage the sequence in LDS
    for (int t = tid; t < nword; t += kThreads) l_ok[t] = a.okbits[p.bits_base + t];
    for (int t = tid; t <= L; t += kThreads) {
      l_dmin[t] = a.p.dmin[p.dmin_base + t];
      l_ws[t] = (KIND == DP_BPP) ? 0. : a.b.ws[p.pos_base + t];
      l_unp[t] = a.b.unp[p.pos_base + t];
      l_seq[t] = (t < L) ? a.b.seq[p.seq_base + t] : 0;
    }
    for (int t = tid; t < nt; t += kThreads) { l_en_o[t] = 0.; l_en_x[t] = 0.; }
    if (tid < 4) l_eh[tid] = 0.;
    if (KIND == DP_SCAN)
      for (int t = tid; t < 3 * (L + 1); t += kThreads) l_post[t] = ELEMDP_NEG_INF;
    __syncthreads();

    pf.lap(0);
    SeqView q;
    q.L = L; q.W = W; q.C = p.C;
    q.seq = l_seq; q.ws = l_ws; q.okbits = l_ok; q.dmin = l_dmin; q.unp = l_unp;
    q.e_stack = a.p.e_stack + p.cell_base; q.e_ext = a.p.e_ext + p.cell_base; q.e_ml = a.p.e_ml + p.cell_base;
    q.e_close = a.p.e_close + p.cell_base; q.e_hp = a.p.e_hp + p.cell_base;
    q.items = a.p.items + p.item_base; q.item_in = a.p.item_in + p.item_base;
    q.by_outer_off = a.p.by_outer_off + p.off_base;
    q.by_inner_off = a.p.by_inner_off + p.off_base; q.by_inner_idx = a.p.by_inner_idx + p.item_base;
    q.by_left_off = a.p.by_left_off + p.off_base; q.by_left_idx = a.p.by_left_idx + p.item_base;
    q.by_right_off = a.p.by_right_off + p.off_base; q.by_right_idx = a.p.by_right_idx + p.item_base;

    TableView Tin, Tout;
    Tin.band = a.band_in + blockIdx.x * a.band_stride;
    Tin.ext = a.ext_in + blockIdx.x * a.ext_stride;
    Tout.band = a.band_out + blockIdx.x * a.band_stride;
    Tout.ext = a.ext_out + blockIdx.x * a.ext_stride;
    Tin.L = Tout.L = L; Tin.W = Tout.W = W; Tin.S = Tout.S = S;

    const Constraint c0{-1, -1, 0};
    GpuSink sink;
    sink.en_ = l_en_o;
    sink.post_[0] = sink.post_[1] = sink.post_[2] = nullptr;

    {
      // ---- schedule of RNAelemScanDP::operator() (motif_scanner.hpp:204-252).  (The fused TRAIN and BPP schedules of round 1 are
      // retired: training runs on the scaled-linear batch pipeline with the log-space batch pipeline as its range fallback, the
      // filter on bpp_kernels.hip; this kernel stays as the scan's range fallback and cross-check.)
      int Ys, Ye;
      if (a.cyk_only) {   // the sum passes ran on the batch pipeline (lin_kernels.hip)
        Ys = a.sc_ys[n]; Ye = a.sc_ye[n];
      } else {
        double* Pys = l_post; double* Pyi = l_post + (L + 1); double* Pye = l_post + 2 * (L + 1);
        sweep_inside<false>(m, q, Tin, c0, no_rss, sc, pf);
        if (tid == 0) { l_zs[0] = part_func(m, Tin, true, true); l_zs[1] = Tin.o(L, m.lay.s00); }
        __syncthreads();
        const double ZL = l_zs[0];
        sink.post_[0] = Pys; sink.post_[1] = Pyi;
        sweep_outside<OUT_SCAN>(m, q, Tin, Tout, ZL, c0, true, true, sink, l_eh, no_rss, sc, pf);
        if (tid == 0) l_zs[4] = (double)last_argmax(Pys, L);
        __syncthreads();
        Ys = (int)l_zs[4];
        for (int t = tid; t < L; t += kThreads) { a.sc_start[p.seq_base + t] = Pys[t]; a.sc_inner[p.seq_base + t] = Pyi[t]; }
        if (a.sc_en) for (int t = tid; t < nt; t += kThreads) a.sc_en[(size_t)n * nt + t] = l_en_o[t];
        const Constraint c1{Ys, -1, 0};
        sweep_inside<true>(m, q, Tin, c1, no_rss, sc, pf);
        if (tid == 0) l_zs[2] = part_func(m, Tin, true, true);
        __syncthreads();
        sink.post_[0] = sink.post_[1] = nullptr; sink.post_[2] = Pye;
        sink.en_ = l_en_x;
        sweep_outside<OUT_END>(m, q, Tin, Tout, l_zs[2], c1, true, true, sink, l_eh, no_rss, sc, pf);
        if (tid == 0) {
          l_zs[5] = (double)last_argmax(Pye, L + 1);
          double tot = ELEMDP_NEG_INF;
          for (int t = 0; t < L; ++t) tot = lse2(tot, Pys[t]);
          a.sc_exist[n] = exp(tot);
        }
        __syncthreads();
        Ye = (int)l_zs[5];
        for (int t = tid; t <= L; t += kThreads) a.sc_end[p.pos_base + t] = Pye[t];
      }
      // Viterbi parse (calc_viterbi_alignment :172-184)
      TraceView R;
      R.ext = a.tr_ext + blockIdx.x * a.ext_stride;
      const Constraint c2{Ys, Ye, 1};
      sweep_cyk(m, q, Tin, R, c2);
      int32_t* path = a.sc_psihat + p.seq_base;
      char* rss = a.sc_rss + p.seq_base;
      for (int t = tid; t < L; t += kThreads) { path[t] = 0; rss[t] = ' '; }
      __syncthreads();
      if (tid == 0) {
        a.sc_ys[n] = Ys; a.sc_ye[n] = Ye;
        const int s0 = Tin.o(L, m.lay.s0m2) < Tin.o(L, m.lay.s0m1) ? m.lay.s0m1 : m.lay.s0m2;
        TraceFrame* stack = reinterpret_cast<TraceFrame*>(a.trace_stack + (size_t)blockIdx.x * a.trace_stack_stride);
        run_trace_back(m, q, Tin, R, c2, L, s0, path, rss, stack, (int)(a.trace_stack_stride * sizeof(int32_t) / sizeof(TraceFrame)));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// deterministic reduction over sequences: one lane per output column, sequences in input order
// partial = [fn, sum_eff, n_used, n_skipped, ENo[nt], ENx[nt], EHo[2], EHx[2]]
// ---------------------------------------------------------------------------------------------
// copies of the items of one sequence in the by_inner / by_left / by_right orders
__global__ __launch_bounds__(kThreads) void k_permute_items(PlanKernelArgs a) {
  const SeqPlan p = a.plans[a.first + blockIdx.y];
  const LoopItem* src = a.p.items + p.item_base;
  const int n = blockIdx.x * kThreads + threadIdx.x;
  if (n >= p.n_items) return;
  a.p.items_inner[p.item_base + n] = src[a.p.by_inner_idx[p.item_base + n]];
  a.p.items_left[p.item_base + n] = src[a.p.by_left_idx[p.item_base + n]];
  a.p.items_right[p.item_base + n] = src[a.p.by_right_idx[p.item_base + n]];
}

// sums of the per-sequence output rows: one workgroup per column, lanes stride over the sequences, then a fixed tree --
// the association depends on nothing but n_seq, so the same rows give the same bits (one lane per column walking all the
// rows took 6 ms per evaluation of 10 000 sequences)
__global__ __launch_bounds__(kThreads) void k_reduce(const double* seq_out, int out_stride, int n_seq, int n_theta,
                                                     double* partial) {
  __shared__ double part[kThreads];
  const int col = blockIdx.x;
  const int src = col == 0 ? 3 : col == 1 ? 5 : (col == 2 || col == 3) ? 4 : 6 + (col - 4);
  double acc = 0.;
  for (int n = threadIdx.x; n < n_seq; n += kThreads) {
    const double v = seq_out[(size_t)n * out_stride + src];
    acc += (col == 2) ? 1. - v : v;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int h = kThreads / 2; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[col] = part[0];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
hipError_t launch_mask(const BatchArrays& b, const SeqPlan* plans, int n_seq, int min_span, bool write_bits, uint32_t* okbits,
                       int32_t* n_canonical, hipStream_t st) {
  if (n_seq <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_mask, dim3(n_seq), dim3(kThreads), 0, st, b, plans, min_span, write_bits ? 1 : 0, okbits, n_canonical);
  return hipGetLastError();
}
hipError_t launch_plan_cells(const PlanKernelArgs& a, int32_t* n_items_out, hipStream_t st) {
  if (a.count <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(n_items_out, 0, sizeof(int32_t) * a.count, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_mask_by_end, dim3((a.nword_max + kThreads - 1) / kThreads, a.count), dim3(kThreads), 0, st, a);
  hipLaunchKernelGGL(k_plan_cells, dim3((a.ncell_max + kPlanTile * kThreads - 1) / (kPlanTile * kThreads), a.count), dim3(kThreads), 0, st, a, n_items_out);
  return hipGetLastError();
}
// the three role lists of a sequence by one workgroup per role with its counters in LDS (k_role_build)?
static bool plan_roles_in_lds(const PlanKernelArgs& a) {
  return a.n_roles == 3 && a.p.items_inner && sizeof(int32_t) * ((size_t)a.ncell_max + 1) <= 150 * 1024 && !getenv("ELEMDP_ROLE_GLOBAL");
}
// .. which then also writes the item copies in the role orders, unless the segments are sorted afterwards
bool plan_copies_fused(const PlanKernelArgs& a) { return plan_roles_in_lds(a) && !a.sort_roles; }
hipError_t launch_permute_items(const PlanKernelArgs& a, hipStream_t st) {
  if (a.count <= 0 || !a.p.items_inner || a.nitems_max <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_permute_items, dim3((a.nitems_max + kThreads - 1) / kThreads, a.count), dim3(kThreads), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_plan_items(const PlanKernelArgs& a, hipStream_t st) {
  if (a.count <= 0) return hipSuccess;
  const dim3 cells((a.ncell_max + 1 + kThreads - 1) / kThreads, a.count), items((a.nitems_max + kThreads - 1) / kThreads, a.count);
  hipLaunchKernelGGL(k_plan_scan, dim3(a.count), dim3(kThreads), 0, st, a, -1);
  hipLaunchKernelGGL(k_plan_fill, dim3((a.ncell_max + kPlanTile * kThreads - 1) / (kPlanTile * kThreads), a.count), dim3(kThreads), 0, st, a);
  if (plan_roles_in_lds(a)) {
    const size_t role_lds = sizeof(int32_t) * ((size_t)a.ncell_max + 1);
    const void* fn = plan_copies_fused(a) ? reinterpret_cast<const void*>(&k_role_build<true>) : reinterpret_cast<const void*>(&k_role_build<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)role_lds);
    if (e != hipSuccess) return e;
    if (plan_copies_fused(a)) hipLaunchKernelGGL(k_role_build<true>, dim3(3, a.count), dim3(kRoleThreads), role_lds, st, a);
    else hipLaunchKernelGGL(k_role_build<false>, dim3(3, a.count), dim3(kRoleThreads), role_lds, st, a);
  } else if (a.n_roles == 3 && a.p.items_inner) {
    hipLaunchKernelGGL(k_role_zero3, cells, dim3(kThreads), 0, st, a);
    if (a.nitems_max > 0) hipLaunchKernelGGL(k_role_count3, items, dim3(kThreads), 0, st, a);
    for (int role = 0; role < 3; ++role) hipLaunchKernelGGL(k_plan_scan, dim3(a.count), dim3(kThreads), 0, st, a, role);
    if (a.nitems_max > 0) hipLaunchKernelGGL(k_role_scatter3, items, dim3(kThreads), 0, st, a);
  } else
  for (int role = 0; role < a.n_roles; ++role) {
    hipLaunchKernelGGL(k_role_zero, cells, dim3(kThreads), 0, st, a, role);
    if (a.nitems_max > 0) hipLaunchKernelGGL(k_role_count, items, dim3(kThreads), 0, st, a, role);
    hipLaunchKernelGGL(k_plan_scan, dim3(a.count), dim3(kThreads), 0, st, a, role);
    if (a.nitems_max > 0) hipLaunchKernelGGL(k_role_scatter, items, dim3(kThreads), 0, st, a, role);
  }
  if (a.sort_roles) return launch_plan_sort(a, st);
  return hipGetLastError();
}
// The scatter leaves the segments of the role lists in arbitrary order.  Sums over them through atomic adds (the scaled-linear
// pipeline) do not care; the gathers of the log-space pipeline and the deterministic mode want a fixed order.
hipError_t launch_plan_sort(const PlanKernelArgs& a, hipStream_t st) {
  if (a.count <= 0 || a.nitems_max <= 0) return hipSuccess;
  const dim3 rows(a.lmax + 1, a.count);
  const bool narrow = a.wmax1 <= 64 && a.nitems_max < (1 << 26);   // 32-bit sort keys (cell << 26 | item)
  for (int role = 0; role < a.n_roles; ++role) {
    if (narrow) {
      hipLaunchKernelGGL((k_role_sort<512, 0, uint32_t>), rows, dim3(kThreads), 0, st, a, role);
      hipLaunchKernelGGL((k_role_sort<2048, 512, uint32_t>), rows, dim3(kThreads), 0, st, a, role);
      hipLaunchKernelGGL((k_role_sort<8192, 2048, uint32_t>), rows, dim3(kThreads), 0, st, a, role);
    } else {
      hipLaunchKernelGGL((k_role_sort<512, 0, unsigned long long>), rows, dim3(kThreads), 0, st, a, role);
      hipLaunchKernelGGL((k_role_sort<2048, 512, unsigned long long>), rows, dim3(kThreads), 0, st, a, role);
      hipLaunchKernelGGL((k_role_sort<8192, 2048, unsigned long long>), rows, dim3(kThreads), 0, st, a, role);
    }
  }
  return hipGetLastError();
}
hipError_t launch_dp(int kind, const DpArgs& a, int n_blocks, hipStream_t st) {
  if (a.n_seq <= 0) return hipSuccess;
  hipError_t e = hipSuccess;
  const size_t lds = (size_t)a.lds.total;
#define ELEMDP_LAUNCH(K)                                                                                          \
  do {                                                                                                            \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dp<K>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                            \
    if (e != hipSuccess) return e;                                                                                \
    hipLaunchKernelGGL(k_dp<K>, dim3(n_blocks), dim3(kThreads), lds, st, a);                                      \
  } while (0)
  switch (kind) {
    case DP_SCAN: ELEMDP_LAUNCH(DP_SCAN); break;
    default: return hipErrorInvalidValue;
  }
#undef ELEMDP_LAUNCH
  return hipGetLastError();
}
hipError_t launch_reduce(const double* seq_out, int out_stride, int n_seq, int n_theta, double* partial, hipStream_t st) {
  const int ncol = 4 + 2 * n_theta + 4;
  hipLaunchKernelGGL(k_reduce, dim3(ncol), dim3(kThreads), 0, st, seq_out, out_stride, n_seq, n_theta, partial);
  return hipGetLastError();
}
const char* dp_kernel_name(int kind) {
  return kind == DP_TRAIN ? "k_dp<0>" : kind == DP_BPP ? "k_dp<1>" : "k_dp<2>";
}

}  // namespace elemdp
