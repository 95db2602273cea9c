// kernels.h -- launch interface between the host engine (engine.cpp) and the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "device_layout.h"
#include "energy_tables.h"
#include "bpp_cand.h"

namespace elemdp {

constexpr int kThreads = 256;  // workgroup size of every kernel (4 waves of 64: one per SIMD of a CU)

// device arrays of one plan set (see SeqPlan for the per-sequence bases)
struct PlanArrays {
  int16_t* dmin = nullptr;
  double* e_stack = nullptr; double* e_ext = nullptr; double* e_ml = nullptr; double* e_close = nullptr; double* e_hp = nullptr;
  int32_t* by_outer_off = nullptr; int32_t* by_inner_off = nullptr; int32_t* by_left_off = nullptr; int32_t* by_right_off = nullptr;
  int32_t* cursor = nullptr;  // scratch, one int per CSR offset entry
  LoopItem* items = nullptr;
  uint8_t* item_in = nullptr;
  int32_t* by_inner_idx = nullptr; int32_t* by_left_idx = nullptr; int32_t* by_right_idx = nullptr;
  // copies of the items in the three secondary orders (one dependent load less per term in the outside sweeps)
  LoopItem* items_inner = nullptr; LoopItem* items_left = nullptr; LoopItem* items_right = nullptr;
};

// static per-batch arrays
struct BatchArrays {
  const uint8_t* seq = nullptr;     // base codes
  const double* ws = nullptr;       // position weights
  const uint8_t* unp = nullptr;     // position may be emitted unpaired
  const int32_t* ndot = nullptr;    // FIX_RSS prefix counts (or null)
};

struct PlanKernelArgs {
  const EnergyTables* et;
  BatchArrays b;
  const uint32_t* okbits;
  uint32_t* okbits_end = nullptr;   // the same mask indexed by (end, span): scratch of the plan builder
  SeqPlan* plans;       // plans[first .. first+count)
  int32_t first, count;
  PlanArrays p;
  int32_t no_ene, min_span, fix_rss;
  int32_t ncell_max = 0, nword_max = 0, nitems_max = 0, lmax = 0;   // largest sequence of the set (grid sizes)
  int32_t wmax1 = 0;     // largest W + 1 of the set
  int32_t n_roles = 3;   // 1: only the by_inner order (plan of the BPP filter)
  int32_t sort_roles = 1;   // sort every segment of the role lists by item index (reproducible summation order of the gathers)
  int32_t count_fast = 0;   // loop_tables_finite and no imposed structure: the count pass takes popcounts (count_interior_by_end)
};

// byte offsets of the dynamic LDS regions of the DP kernels
struct LdsLayout {
  int32_t ints, theta, en_o, en_x, eh, zs, ws, post, okbits, dmin, seq, unp, wave_scr, total;
};

enum DpKind : int { DP_TRAIN = 0, DP_BPP = 1, DP_SCAN = 2 };

struct DpArgs {
  AutomatonLayout lay;            // host-visible copy (launch geometry, LDS sizes)
  const AutomatonLayout* layp;    // the same record in device memory: kernels read it through this pointer
  const int32_t* ints;     // automaton blob (global)
  const double* params;    // ParamBlock followed by theta[n_theta]
  int32_t no_prf, m_min, no_rss;
  int32_t first_pass_only;  // debug: TRAIN stops after the full-terminal outside pass
  int32_t cyk_only;         // SCAN: Ys / Ye are given in sc_ys / sc_ye (batch pipeline); run the Viterbi pass and traceback only
  const SeqPlan* plans;
  const int32_t* order;    // processing order (longest first)
  int32_t n_seq;
  int32_t* counter;        // work queue head
  BatchArrays b;
  const uint32_t* okbits;
  PlanArrays p;
  // table slots, one per workgroup
  double* band_in; double* band_out; double* ext_in; double* ext_out;
  size_t band_stride, ext_stride;  // in doubles
  double* tmp; size_t tmp_stride;  // heavy-sum temporaries: 3 * tmp_stride doubles per slot, tmp_stride = (Lmax+1)*S
  TraceRec* tr_ext;
  // TRAIN: per-sequence results [n][out_stride] = Zo, Zari, Znasi, f, skipped, bpp_eff, ENo[nt], ENx[nt], EHo[2], EHx[2]
  double* seq_out;
  int32_t out_stride;
  // BPP: filtered mask + efficiency
  uint32_t* okbits_out;
  double log_min_bpp;
  double* lnbpp_out;   // optional per-cell ln BPP (debug), indexed by cell_base
  // SCAN outputs (batch offsets: seq_base for start/inner/psihat/rss, pos_base for end)
  double* sc_start; double* sc_end; double* sc_inner; int32_t* sc_psihat; char* sc_rss;
  int32_t* sc_ys; int32_t* sc_ye; double* sc_exist; double* sc_en;  // sc_en: [n][n_theta]
  int32_t* trace_stack; int32_t trace_stack_stride;
  long long* prof;  // optional [n_blocks][8] cycle counters: stage, in-band, in-ext, out-ext, out-band, other
  LdsLayout lds;
};

// arguments of the diagonal-synchronous train pipeline (train_kernels.hip)
struct TrArgs {
  AutomatonLayout lay;
  const AutomatonLayout* layp;
  const int32_t* ints;
  const double* params;
  int32_t no_prf, m_min, no_rss, first_pass_only;
  int32_t lik_ratio;    // --lik-ratio objective (ELEMDP_LIK_RATIO)
  int32_t schedule;     // 0 = the reference's two outside passes, 1 = ari-only + restricted nasi-only (linear)
  int32_t restricted;   // set per launch: this pass sweeps the one-state automaton
  const AutomatonLayout* layp_r; const int32_t* ints_r;
  const SeqPlan* plans;
  const int32_t* grp;   // grp[g] = batch index of the sequence in table slot g
  BatchArrays b;
  const uint32_t* okbits;
  PlanArrays p;
  double* band_in; double* band_out; double* ext_in; double* ext_out;
  size_t band_stride, ext_stride;
  double* tmp; size_t tmp_stride;
  double* seq_out; int32_t out_stride;
  int32_t pass, d;
};
struct BppOut {
  uint32_t* okbits_out;   // filtered pair mask (batch-level, bits_base indexing)
  int32_t* kept;          // kept pairs per sequence (index = position in the plan array)
  double* lnbpp;          // optional ln BPP per cell (cell_base indexing) or null
  double log_min_bpp;
};
// arguments of the linear-semiring BPP filter (bpp_kernels.hip): a chunk of sequences, `plans` = their records with
// cell_base / dmin_base counted from the start of the chunk (seq_base, bits_base: batch level)
// widest band whose Boltzmann weights stay inside the double range with a margin: a GC-rich helix of n stacked pairs weighs
// about e^(5.5 n) (3.4 kcal/mol per stack at kT = 0.616), and a span of W holds at most W / 2 of them -- e^550 at W = 200
// against the limit e^709; wider bands go through the log-space filter (k3_bpp_*)
constexpr int kBppLinMaxSpan = 200;
constexpr int kBppInPlanes = 7 + BC_CLASSES + 1, kBppOutPlanes = 5 + BC_CLASSES;
struct BppLinArgs {
  const EnergyTables* et;
  const EnergyTables* xet;         // the same tables exponentiated (exp_tables): interior loops through loop_weight
  const BppCandTable* cand;        // candidate table of the interior loops (null: the mask walk of round 2)
  int16_t* plist; int32_t* poff;   // pairs of every diagonal of a sequence, ascending ([cell_base + ..], [seq * poff_stride + d]); null: diagonal launches
  int32_t poff_stride, pmax;       // pmax = most canonical pairs of a sequence of the chunk
  int32_t lmax, wmax;              // set by launch_bpp_lin
  unsigned long long* prof;        // debug (ELEMDP_BPP_PROF): cycles of wave 0 of every workgroup per phase [direction][8], or null
  const SeqPlan* plans;
  const uint8_t* seq;
  const uint32_t* okbits;          // canonical pair mask
  int16_t* dmin;                   // [dmin_base + i]: smallest canonical span starting at i (0: none)
  double* xw; size_t xw_stride;    // exp of the five structural terms [term][cell_base + d * (L+1) + i]
  double* tin; double* tout; size_t t_stride;   // band tables [plane][cell_base + d * (L+1) + i]: kBppInPlanes inside / kBppOutPlanes outside planes
  double* lo_in; double* lo_out;   // exterior chains as logarithms [dmin_base + j]
  int32_t no_ene, min_span, m_min, d;
  uint32_t* okbits_out;            // filtered mask (bits_base indexing)
  int32_t* kept;                   // kept pairs per sequence of the chunk
  double* lnbpp;                   // optional ln BPP per candidate [cell_base + i * (W+1) + d], or null
  double log_min_bpp;
};
hipError_t launch_bpp_lin(const BppLinArgs& a, int G, int Lmax, int Wmax, hipStream_t st);
// arguments of the scaled-linear train pipeline (lin_kernels.hip, rules in lin_rules.h)
struct LinArgs {
  AutomatonLayout lay;            // automaton swept by this launch: the full one, or the compact one-state one (S = 1)
  const AutomatonLayout* layp;
  const int32_t* ints;
  const double* params;           // ParamBlock + log theta (lambda, lam_same)
  const double* lin;              // linear parameter block (lin_rules.h: tau, psb, log2 psb, eth)
  int32_t no_prf, m_min, no_rss;
  const SeqPlan* plans;
  const int32_t* grp;             // grp[g] = batch index of the sequence in table slot g
  const SeqPlan* plans_slot;      // the plans of this group in slot order (saves the grp -> plans dependent load), or null
  BatchArrays b;
  const double* ews;              // exp of the position weights
  const uint32_t* okbits;
  PlanArrays p;
  const double* xwc; size_t xwc_stride;   // exp(lambda_k * structural term): [k*5+term][cell]
  const double* xwi; size_t xwi_stride;   // exp(lambda_k * tsc) of the interior-loop items: [k][item]
  double* band_in; double* band_out; double* ext_in; double* ext_out;   // tables swept by this launch
  size_t band_stride, ext_stride;
  double* zs;                     // per slot: mantissas of Z(ari,nasi), Z(ari), Z(nasi), and log2 of the sequence's scale
  double* seq_out; int32_t out_stride;
  int32_t schedule, pass, d;
  int32_t cpb;                    // cells of one block = lanes of a workgroup / lanes per cell of the unary phase
  float rcp_nap, rcp_lane, rcp_cpb, rcp_3cpb;   // 1 / n_ap, 1 / lanes per cell, 1 / cpb, 1 / (3 cpb) for the lane -> (cell, item) splits of the band
                                  // kernels (div_rcp: a reciprocal formed per lane costs ten instructions and a register for the whole kernel)
  int32_t nblk;                   // blocks of cpb consecutive cells a band-kernel workgroup owns (k4_in / k4_out): the context of all
                                  // of them is staged once, the phases then run block by block (set per launch; 0 means 1)
  int32_t* flagged;               // [0] = number of flagged sequences, [1..] = their batch indices
  // scan (sum passes K4 / K5 on this pipeline): start constraint and position-posterior accumulators (batch offsets)
  int32_t lik_ratio;              // --lik-ratio objective (ELEMDP_LIK_RATIO)
  int32_t scan;                   // 1: only Z(ari,nasi) decides the range check
  int32_t* ys; int32_t* ye;       // per batch index: argmax start / end
  double* pos_start; double* pos_inner; double* pos_end; double* exist;
  // scan, Viterbi pass on the batch pipeline: trace tables per slot (same indexing as the band / ext tables), outputs
  TraceRec* tr_ext; int32_t* sc_psihat; char* sc_rss; int32_t* trace_stack; int32_t trace_stack_stride;
  long long* prof;                // optional [16] shader-clock sums per phase (thread 0 of every workgroup), or null
  // rule 2, factorised (lin_rules.h): pair tables [d][i][p] per slot (a_stride doubles each) and the end-indexed pair mask
  // (bits_base indexing, like okbits)
  double* a_in; double* a_out; size_t a_stride;
  const uint32_t* okbits_end;
  int32_t lmax, nword_max;        // longest sequence of the launch / most pair-mask words of a sequence (LDS sizing)
  int32_t wmax;                   // largest span of the launch (sizes the position window staged in LDS)
  int32_t n_stage;                // ints of the automaton blob staged in LDS: n_ints (whole blob) or n_small
  int32_t dbg;                    // timing experiments only: bit 0 skip split sums, 1 skip item sums, 2 skip the unary phase
  int32_t ext_ring;               // exterior-chain kernels keep the chain's last rows in an LDS ring (small groups only)
  int32_t ext_block;              // steps of the inside exterior chain whose pair sums are formed side by side (4 where every pair
                                  // spans >= 5 positions -- the default mask --, else 1)
  int32_t n_lin;                  // doubles of the linear parameter block the band kernels stage (with or without the weight tables)
  int32_t fast;                   // train: table-driven unary phases (lin_fast.h); the host clears it where they do not apply
  int32_t det;                    // train: deterministic reductions -- every heavy sum of a workgroup gets its adds from ONE wave (pairs of
                                  // a cell in one wave: det_sh; tuples dealt to the waves by target: AutomatonLayout::qd_*), the
                                  // statistics one copy per wave, one row of counts per (sequence, block):
                                  // det_rows[n][det_nslot][out_stride], summed in order by k4_combine
  int32_t cyk_compact;            // scan, Viterbi pass: the table in the compact layout (TableView::ldm / stm), set by launch_cyk_group
  int32_t det_sh;                 // log2 of the lanes a cell's pairs take in the pair phases of the deterministic mode (a power of two
                                  // >= n_ap, so that no cell straddles two waves); -1: more than 64 pairs, one wave does the phase
  double* det_rows; int32_t det_nslot;
};
struct LinWeightArgs {
  const LoopItem* items_inner; const LoopItem* items_left; const LoopItem* items_right;   // (may be null)
  const double* e_stack; const double* e_ext; const double* e_ml; const double* e_close; const double* e_hp;
  const LoopItem* items;
  size_t n_cells, n_items;       // cells of the batch (= stride of the planes of xwc), items
  size_t cell_first = 0, cell_count = 0;   // cells to compute (count 0: all) -- an evaluation of a range of the batch
  const double* params;
  double* xwc; double* xwi;
};
hipError_t launch_lin_weights(const LinWeightArgs& a, hipStream_t st);
// one whole train evaluation of a group; `full` sweeps the pattern automaton, `compact` (schedule 1) the one-state
// automaton of the no-motif pass over the compact tables
// scan: phase 0 = inside + outside with start / inner posteriors and argmax start; phase 1 = the same constrained to that
// start with end posteriors and argmax end (RNAelemScanDP::operator(), motif_scanner.hpp:186-202)
hipError_t launch_lin_scan_group(const LinArgs& full, int G, int Lmax, int Wmax, int phase, hipStream_t st);
// scan: Viterbi parse (max-plus CYK with trace records, then traceback) of a group; uses band_in / ext_in as the CYK table
hipError_t launch_cyk_group(const LinArgs& full, int G, int Lmax, int Wmax, hipStream_t st);
hipError_t launch_lin_group(const LinArgs& full, int G, int Lmax, int Wmax, bool first_pass_only, hipStream_t st);
hipError_t launch_bpp_group(const TrArgs& base, const BppOut& o, int G, int Lmax, int Wmax, hipStream_t st);
hipError_t launch_train_group(const TrArgs& base, int G, int Lmax, int Wmax, hipStream_t st);

hipError_t launch_mask(const BatchArrays& b, const SeqPlan* plans, int n_seq, int min_span, bool write_bits, uint32_t* okbits,
                       int32_t* n_canonical, hipStream_t st);
hipError_t launch_plan_cells(const PlanKernelArgs& a, int32_t* n_items_out, hipStream_t st);
hipError_t launch_plan_items(const PlanKernelArgs& a, hipStream_t st);
hipError_t launch_plan_sort(const PlanKernelArgs& a, hipStream_t st);   // the sort of launch_plan_items alone (sort_roles = 0 before)
hipError_t launch_permute_items(const PlanKernelArgs& a, hipStream_t st);
bool plan_copies_fused(const PlanKernelArgs& a);   // launch_plan_items has written the item copies already (no launch_permute_items)
hipError_t launch_dp(int kind, const DpArgs& a, int n_blocks, hipStream_t st);
hipError_t launch_reduce(const double* seq_out, int out_stride, int n_seq, int n_theta, double* partial, hipStream_t st);
const char* dp_kernel_name(int kind);

}  // namespace elemdp
