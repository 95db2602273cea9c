// device_layout.h -- PODs shared by the host engine and the HIP kernels.
//
// Naming follows the reference's domain: structural states P,E,M,B,1,2,L(,O) (energy_model.hpp:58-70),
// interval states s=(l,r) of the pattern automaton (profile_hmm.hpp:21-29), rules numbered as in
// SURVEY.md Appendix A.
#pragma once
#include <cstdint>

namespace elemdp {

enum StructState : int { ST_P = 0, ST_E = 1, ST_M = 2, ST_B = 3, ST_1 = 4, ST_2 = 5, ST_L = 6, ST_O = 7 };
constexpr int kNumBandStates = 7;  // P,E,M,B,1,2,L live in the banded tables; O is the exterior prefix

// Transition codes stored in CYK trace records (values follow energy_model.hpp:72-91)
enum TransType : int {
  TT_E_H = 0, TT_P_E, TT_P_P, TT_O_O, TT_O_OP, TT_E_P, TT_E_M, TT_M_M, TT_M_B, TT_B_12, TT_1_B, TT_1_2, TT_2_2,
  TT_2_P, TT_L_L
};

// ---- flattened pattern automaton ------------------------------------------------------------
// All lists are CSR over interval-state ids: entries of state s are [off[s], off[s+1]).
// "fwd" lists enumerate children of a parent (inside direction), "rev" lists enumerate parents
// of a child (outside direction, gather form).  Offsets below index ModelBlob::ints.
struct AutomatonLayout {
  int32_t S;        // interval states (= stride of the state axis of every table)
  int32_t n_active; // states [0, n_active) are swept: S, or 1 in the restricted automaton of the no-motif pass
  int32_t n_front;  // states [0, n_front) are the only ones that can take part in a bifurcation (non-zero in planes B, 1, 2
                    // of a complete parse); the split-sum kernels stage only this prefix of a row.  S without pruning.
  int32_t shadow;   // id of the shadow copy of state (0,0) (Automaton::flatten), or -1
  int32_t M;        // pattern nodes incl. 'z' and 'o'
  int32_t n_theta;  // total theta entries (n_param - 2)
  int32_t n_rows;   // theta rows
  int32_t s00, s0m1, s0m2;  // ids of (0,0), (0,M-1), (0,M-2)  (terminal states, motif_trainer.hpp:100-112)
  // per-state attributes
  int32_t st_l, st_r;        // node indices
  int32_t st_is_loop;        // reachable as loop (profile_hmm.hpp:380-383)
  int32_t st_row_r;          // theta row of node r (-1: none)
  int32_t st_row_l;          // theta row of node l (-1: none)
  int32_t st_pair_r;         // node r is ')'
  int32_t st_w_r, st_w_l;    // position weight applies to node r / l ('.', '(' or ')': motif_model.hpp:131-134)
  int32_t st_lam;            // 0 if l==r else 1 (motif_model.hpp:117-121)
  int32_t st_ref;            // id of the state in the reference's order (profile_hmm.hpp:228-260); tables are exported in it
  int32_t row_off;           // n_rows+1 offsets of theta rows in the parameter vector
  // forward lists: (child, tau_flag) pairs, 2 ints per entry
  int32_t right_off, right_ent;  // loop_right_trans  (profile_hmm.hpp:389-401)
  int32_t left_off, left_ent;    // loop_left_trans   (:403-415)
  int32_t pair_off, pair_ent;    // pair_trans        (:417-448)
  // splits s -> (s1=(l,h), s2=(h,r)), 2 ints per entry (motif_model.hpp:368-381, 315-327)
  int32_t split_off, split_ent;
  // interior-loop quadruples grouped by parent s: (s1, s2, s3), 3 ints (profile_hmm.hpp:451-463)
  int32_t quad_off, quad_ent;
  // reverse lists (grouped by CHILD): (parent, tau_flag)
  int32_t rright_off, rright_ent;
  int32_t rleft_off, rleft_ent;
  int32_t rpair_off, rpair_ent;
  // splits grouped by s1 (the "1" / prefix child): (parent, s2); by s2 (the "2" / pair-part child): (parent, s1)
  int32_t split1_off, split1_ent;
  int32_t split2_off, split2_ent;
  // quadruples grouped by s1 (P child): (s, s2, s3); by s2 (left loop): (s, s1, s3); by s3: (s, s1, s2)
  int32_t quad1_off, quad1_ent;
  int32_t quad2_off, quad2_ent;
  int32_t quad3_off, quad3_ent;
  // target (= grouping) state of every flat entry of the seven tuple lists above, for lane-per-tuple gathers
  int32_t split_tgt, split1_tgt, split2_tgt, quad_tgt, quad1_tgt, quad2_tgt, quad3_tgt;
  int32_t n_split, n_quad;  // entries per split* list / per quad* list
  // Rule 2 through the pair-sparse factorisation (lin_rules.h): "pairs" p = (s1, t) of a state s1 of plane 1 and a state t
  // of plane 2 with s1.r == t.l, closed under the right-emission predecessors of t.  Per pair: ap_s1, ap_t, ap_tgt (the
  // parent s of the split (s; s1, t), or -1).  ap_chain: CSR by pair p -> (p', tau flag) with t' in right(t) (inside
  // direction); ap_rchain: the same entries by p' -> (p, tau flag); ap_by_s1 / ap_by_t: CSR by state -> pair ids.
  int32_t n_ap;
  int32_t ap_s1, ap_t, ap_tgt;
  int32_t ap_chain_off, ap_chain_ent, ap_rchain_off, ap_rchain_ent;
  int32_t ap_by_s1_off, ap_by_s1_ent, ap_by_t_off, ap_by_t_ent;
  // Compact band tables of the scaled-linear pipeline (lin_rules.h): plane e keeps one row of tab_rs[e] doubles per cell with
  // one column per interval state that is USEFUL in that plane (Automaton::liveness; every state without pruning):
  // ints[tab_cmap + e * S + s] = column of state s in plane e, or -1 (the entry is 0 in every complete parse and is neither
  // stored nor read).  tab_cs[e] = sum of the strides of the planes before e, tab_row = sum of all seven; the plane of a
  // sequence starts at tab_cs[e] * (W+1) * (L+1).  Strides are padded (multiples of 8 doubles) so that rows start on 64-byte
  // lines.  ap_rs: row stride of the pair tables of the factorised rule 2 (>= n_ap).
  int32_t tab_cmap;
  int32_t tab_rs[7], tab_cs[7], tab_row;
  // tab_cell: the seven rows of a cell lie side by side in one record of tab_row doubles (entry = cell * tab_row + tab_cs[e] + column)
  // instead of plane after plane (entry = tab_cs[e] * cells + cell * tab_rs[e] + column): TableView::set_compact
  int32_t tab_cell;
  int32_t ap_rs;
  // ---- Table-driven train kernels (lin_fast.h, k4_in / k4_out with FAST).  They stage only their own FAST BLOB -- ints
  // [fb_in, fb_in + fb_in_n) resp. [fb_out, fb_out + fb_out_n) behind n_ints -- instead of the generic lists:
  //   * one PROGRAM of kFastW ints per interval state (fp_in / fp_out): the state's own columns, and per unary transition the
  //     operand columns and the id of the transition = its position in the forward list (right / pair / left), which indexes
  //     the per-evaluation weight tables WR[id][base], WL[id][base], WP[id][pair type] behind the linear parameter block
  //     (lin_params.h) and the static attribute tables fe_r (2 ints per right transition) / fe_p (3 per pair transition);
  //   * one PAIR RECORD of 8 ints per pair (s1, t) of the factorised rule 2 (fpr_in / fpr_out): {c1 | cP << 8 | tgt << 16 |
  //     flags << 24 (bit 0 lambda class of t, 1 position weight of t's r-node, 2 t is the shadow state); s1 | t << 8 |
  //     n_chain << 16 | n_rchain << 20; 3 chain entries pc | id << 8 (tail step from pair pc, right transition id); 3 rchain
  //     entries pp | id << 8} with c1 / cP = column of s1 in plane 1 / of t in plane P, tgt = 0xff for none;
  //     tgt, s1 and t are LIVE INDICES (position among the states that have a column at all, f_live_*): the heavy sums of a cell
  //     are n_lane wide in the table-driven kernels, not S; the targets of the fast tuple records (fqc_*) likewise;
  //   * the column records of the interior-loop tuples (fqc_in; fqc_out = three lists of n_quad, see qc_* below).
  // fp_ok = 0: a list is longer than kFastR / kFastP / kFastL (or a column index does not fit a byte); the kernels then
  // run the generic rule code.
  int32_t fp_ok, fb_in, fb_in_n, fb_out, fb_out_n;
  int32_t st_live, st_li;   // the same list (n_lane ints) and its inverse (S ints, -1: no column anywhere) in the SMALL part of the blob
  int32_t n_lane;     // interval states with a column in at least one plane: the unary phases give a lane to each of them per cell
  int32_t f_live_in, f_live_out;   // their ids (n_lane ints, one copy in either fast blob)
  int32_t fp_max_p;   // longest pair list (forward or reverse) of a state: the kernels unroll 2 or kFastP slots
  int32_t fp_in, fqc_in, fpr_in, fp_out, fe_r, fe_p, fqc_out, fpr_out;
  // Scan passes on the table-driven kernels: what a scanner functor tests on the nodes of an emitting transition
  // (motif_scanner.hpp:546-573, 594-622, 715-747) as one word of ScanFlag bits per forward transition, a copy in either fast
  // blob: [fs_*, + n_wr) right, [+ n_wr, + n_wr + n_wl) left, [+ n_wr + n_wl, ...) pair transitions.
  int32_t fs_in, fs_out;
  int32_t n_wr, n_wp, n_wl;                     // transitions per forward list = rows of the weight tables
  int32_t lin_wr, lin_wl, lin_wp, lin_total;    // offsets (doubles) of the weight tables in the linear block; its length
  // Column records of the interior-loop tuples (rule 6c), 2 ints each, in the tuple-list runs of the blob: the item sums hold
  // one item record per lane and read the tuples' operand columns from here instead of deriving them from the state ids.
  // {columns of the first, second, third operand and of the target's own entry, one byte each; target state | flags << 16
  // (bit 0 lambda class of the rule's parent, 1 the parent is the shadow state, 2 a column is missing: the tuple is dead)}
  //   qc_in  (quad order):  P(inner) [s1], L(left) [s2], L(right) [s3], -; target = parent E state
  //   qc_out1 (quad1 order, target = the inner pair's P state): out E [par], L [s2], L [s3], own = P [target]
  //   qc_out2 / qc_out3 (quad2 / quad3 order, target = a loop's L state): out E [par], P [s1], L [other loop], own = L [target]
  int32_t qc_in, qc_out1, qc_out2, qc_out3;
  // Deterministic mode (LinArgs::det): the tuples of a list dealt to the four waves of a band-kernel workgroup BY TARGET (target
  // index & 3), so that every heavy sum is added to by one wave only -- its adds then reach LDS in program order, lanes of one
  // instruction in lane order.  Per list 5 offsets (into the ids that follow them) + n_quad tuple ids: qd_in for qc_in, qd_out for the
  // three lists at qc_out1 (5 + n_quad ints each); fqd_in / fqd_out: the same for the fast copies (targets = live indices).
  int32_t qd_in, qd_out, fqd_in, fqd_out;
  int32_t n_small;  // the first n_small ints (per-state attributes, unary lists) are staged in LDS;
                    // the tuple lists behind them are read from global memory (ModelView::big)
  int32_t big_in_end;  // the tuple lists behind n_small come in two runs: [n_small, big_in_end) = lists of the inside
                       // direction (split, quad + targets), [big_in_end, n_ints) = lists of the outside direction
  int32_t n_ints;   // total length of the int blob
};

// nodes of the transition parent (pl, pr) -> child (cl, cr) of a pattern with nodes 0 .. M-1
enum ScanFlag : int {
  SF_SL = 1,    // pl == 0 && cl == 1:       the left emission starts the motif
  SF_SR = 2,    // cr == 0 && pr == 1:       the right emission starts the motif
  SF_IL = 4,    // cl != 0 && cl != M-1:     the left emission lies inside the motif
  SF_IR = 8,    // pr != 0 && pr != M-1:     the right emission lies inside the motif
  SF_EL = 16,   // pl == M-2 && cl == M-1:   the left emission is the first position behind the motif
  SF_ER = 32,   // cr == M-2 && pr == M-1:   the right emission is the first position behind the motif
  SF_PM2 = 64   // pr == M-2:                the motif ends with the sequence if the emission is its last position
};
constexpr int kFastW = 16;                        // ints per unary program
constexpr int kFastR = 3, kFastP = 3, kFastL = 2;  // most right / pair / left transitions of a state the programs (and the kernels) hold

// Per-evaluation parameters (changes every optimizer step)
struct ParamBlock {
  double lambda[2];
  double log_tau;
  int32_t lam_same;  // lambda[0]==lambda[1] bitwise-equal as doubles (motif_trainer.hpp:380-381 quirk)
  int32_t pad;
  // followed in memory by theta[n_theta] (log-probabilities; rows per row_off)
};

// ---- per-sequence plan (parameter independent, built once per batch on the GPU) -------------
struct SeqPlan {
  int32_t L, W, C;
  int32_t positive;     // last quality == 0  <=>  ws[L] == -inf  ("has motif", motif_model.hpp:69)
  // batch-level (static) arrays
  int64_t seq_base;     // base codes (L entries)
  int64_t pos_base;     // per-position arrays ws / unp (L+1 entries)
  int64_t bits_base;    // pair-mask words (ceil((L+1)*(W+1)/32) words)
  // plan-level arrays (relative to the plan set the sequence belongs to)
  int64_t dmin_base;    // L+1 entries
  int64_t cell_base;    // per-cell terms ((L+1)*(W+1) entries)
  int64_t off_base;     // CSR offsets ((L+1)*(W+1)+1 entries)
  int64_t item_base;    // interior-loop items
  int32_t n_items;      // outside set; members of the inside set are flagged
  int32_t n_canonical;  // base pairs possible by sequence alone (denominator of bpp_eff)
  double bpp_eff;       // kept / possible pairs (energy_model.hpp:265)
  int32_t index;        // position of the sequence in the batch (copies of the plan array kept in processing order)
  int32_t pad_;
};

// interior-loop item: outer cell E(i,j), inner pair P(k,l), tsc = loop_energy(i-1,j,k,l-1)
struct LoopItem {
  double tsc;
  int16_t i, j, k, l;
};

// CYK trace record (motif_scanner.hpp:51-59): child cell (k,l), transition type t (< 0 = leaf),
// child structural state e1 and child interval state s1
struct TraceRec {
  int16_t k, l;
  int8_t t, e1;
  int16_t s1;
};

}  // namespace elemdp
