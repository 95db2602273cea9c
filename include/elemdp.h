/* elemdp.h -- C ABI of libelemdp.so, the MI355X-native inside/outside/CYK engine.
 *
 * The reference (iyak/RNAelem @ 2024_08_07) has no FFI: its de-facto operator interface for this
 * path is C++ duck typing,
 *     int  RNAelemTrainer::operator()(V const& x, double& fn, V& gr)   RNAelem/motif_trainer.hpp:595
 *     void RNAelemScanner::scan(RNAelem& model)                        RNAelem/motif_scanner.hpp:938
 * consumed by Lbfgsb::minimize / Adam::minimize (RNAelem/optimizer.hpp:146, :298) and main()
 * (RNAelem/main.cpp:47-130).  This header is the boundary a maintainer would bind instead
 * (INTEGRATION.md shows the C++ shim that drops into motif_trainer.hpp / motif_scanner.hpp).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success, a negative
 * ELEMDP_E* code otherwise (elemdp_last_error() gives the message); no exceptions cross the
 * boundary.  Caller owns every buffer it passes; inputs are copied during the call.  One handle =
 * one GPU = one caller thread.  The library has NO CPU fallback: without a usable HIP device
 * elemdp_create fails with ELEMDP_ENODEV.
 */
#ifndef ELEMDP_H
#define ELEMDP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ELEMDP_ABI_VERSION 1

enum { /* status codes */
  ELEMDP_OK = 0,
  ELEMDP_EINVAL = -1,  /* bad argument / malformed pattern or parameter text */
  ELEMDP_ENODEV = -2,  /* no HIP device */
  ELEMDP_EHIP = -3,    /* HIP runtime error */
  ELEMDP_ESTATE = -4,  /* call order (e.g. train_eval before load_batch) */
  ELEMDP_ENOMEM = -5,
};

enum { /* elemdp_model_desc.flags */
  ELEMDP_NO_RSS = 1 << 0,        /* --no-rss       RNAelem/application.hpp:259, motif_model.hpp:171-206 */
  ELEMDP_NO_PROFILE = 1 << 1,    /* --no-profile   RNAelem/application.hpp:265 */
  ELEMDP_NO_ENERGY = 1 << 2,     /* --no-energy    RNAelem/application.hpp:271 */
  ELEMDP_THETA_SOFTMAX = 1 << 3, /* --theta-softmax RNAelem/application.hpp:289 */
  ELEMDP_LIK_RATIO = 1 << 4,     /* --lik-ratio: a sequence without motif contributes Z(ari) - Z(ari,nasi) and the statistics of
                                    those two terminal sets (RNAelem/motif_trainer.hpp:156-202, --no-shuffle branch) */
  /* runtime forms of the reference's compile-time test switches (RNAelem/const_options.hpp:12-24) */
  ELEMDP_DBG_FIX_RSS = 1 << 9,   /* structure fixed per sequence (elemdp_load_batch `fix_rss`) */
  ELEMDP_DBG_NO_TURN = 1 << 10,  /* hairpins of any size */
};

/* Model description == what main.cpp:89-101 / RNAelemReader::read_model (motif_io.hpp:118-262)
 * put into an `RNAelem` object before the optimizer / scanner is started. */
typedef struct {
  const char* pattern;      /* search pattern, e.g. "((.*.))"         --motif-pattern        */
  const char* energy_param; /* ViennaRNA-2.0 parameter TEXT, or NULL / "~T2004~" / "~A2007~"
                               for the shipped tables (data dir: elemdp_set_data_dir)      */
  int32_t max_span;         /* --max-span           (default 50)                            */
  int32_t max_iloop;        /* --max-internal-loop  (default 30)                            */
  double min_bpp;           /* --min-bpp            (default 1e-4; 0 = no BPP filter)       */
  double tau;               /* --tau                (default 0.1)                           */
  int32_t flags;            /* ELEMDP_* bits                                                */
  int32_t device;           /* HIP device ordinal; -1 = current                             */
} elemdp_model_desc;

typedef struct elemdp_handle elemdp_handle;

const char* elemdp_last_error(void);
int elemdp_abi_version(void);
/* directory holding turner2004.elempar / andronescu2007.elempar (default: next to the library) */
int elemdp_set_data_dir(const char* dir);

/* Builds the pattern automaton (RNAelem::set_motif_pattern, motif_model.hpp:80-97) and the energy
 * tables (EnergyModel::set_param_file, energy_model.hpp:153-161), uploads them, creates streams. */
int elemdp_create(const elemdp_model_desc* desc, elemdp_handle** out);
int elemdp_destroy(elemdp_handle* h);

/* Sizes: n_param = #theta entries + 2 (pack_params order, motif_model.hpp:147-157);
 * n_state = S interval states; n_node = M pattern nodes (incl. 'z' and 'o'). */
int elemdp_n_param(const elemdp_handle* h);
int elemdp_n_state(const elemdp_handle* h);
int elemdp_n_node(const elemdp_handle* h);
/* x0 exactly as the reference CLI builds it: uniform log-probability rows (profile_hmm.hpp:286-313;
 * zeros when theta-softmax), then lambda_init twice (motif_trainer.hpp:565). */
int elemdp_initial_params(const elemdp_handle* h, double lambda_init, double* x, int32_t n_param);
/* JSON description of the automaton (states, transition lists) for inspection / host-logic tests. */
int elemdp_describe(const elemdp_handle* h, char* buf, int32_t cap);

/* Engine knobs (not part of the reference interface).  Unknown keys are ELEMDP_EARG.
 *   evaluation
 *     "pipeline"        4 (default): scaled-linear batch pipeline, which hands sequences outside the double range to the log-space
 *                       one; 3: log-space batch pipeline for everything.  (2, the fused kernel of round 1, is retired.)
 *     "schedule"        1 (default): ONE outside sweep for both passes of motif_trainer.hpp:209-225 -- "has motif" terminals on
 *                       the pattern's states, "no motif" terminal on a shadow copy of state (0,0); 0: the reference's two sweeps
 *     "fast"            1 (default): table-driven band kernels (per-state programs, weight tables, cell records); 0: generic rule code
 *     "nblk"            blocks of cells a band-kernel workgroup owns: 0 (default) = chosen per launch, n = n wherever they fit
 *     "deterministic"   1: bit-identical repeats of elemdp_train_eval (fixed summation order, as the reference at --thread 1); slower
 *     "eval_first", "eval_count"   a train evaluation covers the records [first, first + count) of the resident batch only
 *                       (count 0 = all; reset by elemdp_load_batch).  Refused (ELEMDP_EARG) for a streamed batch and for pipeline 3
 *     "prune"           1 (default): transition lists without what cannot occur in a complete parse; 0: the reference's complete lists
 *     "first_pass_only" debug: stop a train evaluation after the first outside pass
 *   batches
 *     "max_resident"    most sequences kept resident at a time (0 = as many as the device memory holds): a larger batch is STREAMED --
 *                       elemdp_train_eval / elemdp_scan run it in chunks of that size, the BPP filter + plan of chunk k+1 built on a
 *                       second inner engine and host thread while chunk k is evaluated, partial sums added in chunk order; set
 *                       before elemdp_load_batch
 *     "group"           sequences swept in lockstep (0 = as many as fit); "group_streams": groups evaluated concurrently (default 2);
 *                       "two_streams": second outside pass of schedule 0 on a second stream (default 1); "slots": table slots
 *     "keep_lnbpp"      keep ln BPP of the filter for elemdp_batch_pairs; "bpp_log": 1 = log-space BPP filter for every band
 *     "sorted_plan"     1: role lists of the plan sorted per cell (reproducible summation order of the log-space pipeline)
 *     "row_pad"         padding of the compact table rows in doubles (default 1 = none; 8 = rows start on 64-byte lines)
 *     "cell_major"      1 = the seven rows of a cell side by side in one record instead of plane after plane (default 0)
 *   measurement / tests
 *     "profile"         in-kernel phase clocks for elemdp_debug_profile; "dbg": switch phases off (results invalid);
 *     "poison"          1: every table is filled with NaN before an evaluation (an unmasked read of an entry nobody stored shows) */
int elemdp_set_option(elemdp_handle* h, const char* key, double value);

/* Replaces the resident batch (== FastqReader contents, fastq_io.hpp:64-108):
 *   seq_codes : concatenated base codes N,A,C,G,U -> 0..4 (bio_sequence.hpp:28-39)
 *   seq_off   : n_seq+1 offsets;   qual : char-33 values, L+1 per sequence;   qual_off likewise
 *   fix_rss   : NULL, or concatenated dot-bracket strings (seq_off indexing) with ELEMDP_DBG_FIX_RSS
 * Runs the parameter-independent part once on the GPU and keeps it resident:
 * the BPP filter (EnergyModel::set_seq .. fill_bpp_tables, energy_model.hpp:211-276) and the
 * structural energy terms of every admissible rule (energy_param.hpp:686-795). */
int elemdp_load_batch(elemdp_handle* h, const uint8_t* seq_codes, const int32_t* seq_off, const uint8_t* qual,
                      const int32_t* qual_off, const char* fix_rss, int32_t n_seq);
/* per-sequence results of the BPP filter: bpp_eff[n_seq] (energy_model.hpp:265) */
int elemdp_batch_bpp_eff(elemdp_handle* h, double* bpp_eff, int32_t n_seq);
/* kept[(L+1)*(W+1)] (index i*(W+1)+d) of one sequence after the filter; lnbpp may be NULL */
int elemdp_batch_pairs(elemdp_handle* h, int32_t seq_index, uint8_t* kept, double* lnbpp, int32_t cap);

/* == RNAelemTrainer::operator()(x, fn, gr) over the whole resident batch with --no-shuffle
 * (motif_trainer.hpp:595-633 + RNAelemTrainDP::operator() :124-272).  fn/gr are the UNREGULARISED
 * sums (the optimizer adds rho*x^2/2, optimizer.hpp:246-260).  sum_eff = sum of bpp_eff over used
 * sequences (:227); n_skipped = sequences with non-finite Z (:211-215). */
int elemdp_train_eval(elemdp_handle* h, const double* x, int32_t n_param, double* fn, double* gr,
                      double* sum_eff, int32_t* n_skipped);

/* Multi-GPU form: the same evaluation, but stops before the cross-rank sum.  `partial` (device or
 * host pointer, elemdp_partial_len(h) doubles) receives this rank's
 *   [fn, sum_eff, n_used, n_skipped, ENo[n_theta], ENx[n_theta], EHo[2], EHx[2]]
 * The caller all-reduces (sum) it over RCCL -- the MI355X replacement of the reference's
 * file-based array-job sum (motif_array_trainer.hpp:20-58) -- and then calls
 * elemdp_train_finish on the reduced vector (host pointer) to obtain fn / gr. */
int elemdp_partial_len(const elemdp_handle* h);
int elemdp_train_partial(elemdp_handle* h, const double* x, int32_t n_param, void* partial, int32_t partial_is_device);
int elemdp_train_finish(elemdp_handle* h, const double* reduced, double* fn, double* gr, double* sum_eff,
                        int32_t* n_skipped);
/* Host-only: tells the handle which x a following elemdp_train_finish refers to (needed for the
 * softmax chain rule, motif_trainer.hpp:251-261) when elemdp_train_partial ran in another handle. */
int elemdp_set_finish_params(elemdp_handle* h, const double* x, int32_t n_param);

/* In-library collective for hosts without their own (the reference binary with INTEGRATION.md's shim): one process (or
 * thread) per GPU, one handle each.  Rank 0 obtains an id with elemdp_comm_unique_id (128 bytes, ncclUniqueId) and hands
 * it to the other ranks by whatever means the host has (a file, MPI, a socket); every rank then calls elemdp_comm_init on
 * its handle.  From then on elemdp_train_eval all-reduces (sum, fp64, elemdp_partial_len doubles) the partial vector over
 * RCCL / xGMI on the engine's stream before it finishes fn / gr, so every rank returns the values of the WHOLE batch --
 * the replacement of the array job + result files of motif_array_trainer.hpp:20-58 (submit_array_job / collect_fn_gr_eff).
 * A rank whose share of the batch is empty calls elemdp_train_eval without a batch: it contributes zeros.
 * librccl.so is loaded on the first call (dlopen); without it these return ELEMDP_ENODEV. */
#define ELEMDP_COMM_ID_BYTES 128
int elemdp_comm_unique_id(void* id_out);
int elemdp_comm_init(elemdp_handle* h, int32_t rank, int32_t world, const void* id);
int elemdp_comm_destroy(elemdp_handle* h);

/* per-sequence diagnostics of the last train evaluation: 5 doubles per sequence
 * [Z(ari,nasi), Z(ari), Z(nasi), f_n, skipped] (motif_trainer.hpp:108-112, 204-227) */
int elemdp_train_seq_stats(elemdp_handle* h, double* out, int32_t n_seq);
/* debug: tables of ONE sequence after a train evaluation of a batch holding only that sequence:
 * inside_o/outside_o [(L+1)*S]; inside/outside [(L+1)*(W+1)*7*S] in the reference's index order
 * [i][d][e][s] (motif_trainer.hpp:62-65); outside = the first (full-terminal) pass.  Any may be NULL.
 * ENo/ENx [n_theta], EH [4] = EHo,EHx. */
int elemdp_debug_tables(elemdp_handle* h, double* inside, double* outside, double* inside_o, double* outside_o,
                        double* ENo, double* ENx, double* EH);

/* == RNAelemScanner::scan (motif_scanner.hpp:938-949, per-sequence worker :215-260), input order
 * preserved.  Caller-provided arrays use the batch offsets: start/inner at seq_off[n] (L values),
 * end at qual_off[n] (L+1 values), psihat/rss at seq_off[n]; per-sequence scalars indexed by n. */
typedef struct {
  double* start;      /* log P(motif starts at p)                  "start:"  */
  double* end;        /* log P(motif ends at p | start = Ys)       "end:"    */
  double* inner;      /* log P(p inside motif)                     "inner:"  */
  int32_t* psihat;    /* CYK motif node per position               "psihat:" */
  char* rss;          /* CYK structure letters O L R H B I M       "rss:"    */
  int32_t* ys;        /* argmax start (last maximum, util.hpp:232) "motif region:" */
  int32_t* ye;
  double* exist_prob; /* exp(logsumexp(start))                     "exist prob:" */
  double* en;         /* n_theta expected emission counts summed over the batch ("E[N]:"), may be NULL */
} elemdp_scan_out;
int elemdp_scan(elemdp_handle* h, const double* x, int32_t n_param, elemdp_scan_out* out);

/* timing of the last train evaluation, measured with HIP events on the engine's stream:
 * ms[0] = whole evaluation, ms[1] = the DP pipeline only (all kernels of the inside/outside sweeps),
 * ms[2] = number of sequences the scaled-linear pipeline handed to the log-space pipeline (range check) */
int elemdp_last_timing(elemdp_handle* h, double* ms, int32_t n);
/* debug: summed shader-clock cycles per phase of the last train evaluation when option "profile" = 1:
 * pipeline 2: [stage, inside band, inside exterior, outside exterior, outside band, queue/other];
 * pipeline 4 (16 values): k4_in [setup, stage split operands, split products, item sums, unary phase],
 * k4_out [setup, stage, products, items inner, items left, items right, unary phase, statistics flush] */
int elemdp_debug_profile(elemdp_handle* h, double* cycles, int32_t n);
/* Host only: the shuffled negative of one sequence as `elem train` (without --no-shuffle) generates it per iteration:
 * k-let preserving Euler-tour shuffle (uShuffle, Jiang et al. 2007; RNAelem/ushuffle/ushuffle.c:139-290) driven by the C
 * library's srand(seed) / rand() exactly as RNAelem/motif_trainer.hpp:145-152 does (seed = occurrences of the first base
 * + iteration count).  codes / out: L base codes.  Not thread safe (rand() is process global). */
int elemdp_kmer_shuffle(const uint8_t* codes, int32_t L, int32_t k, int32_t iter_cnt, uint8_t* out);
/* Host only: the order in which `elem train --batch-size N` reads the records in epoch `seed`+1: the permutation that
 * std::shuffle(first, last, std::mt19937(seed)) applies to an array of n elements (FastqReader::shuffle,
 * RNAelem/fastq_io.hpp:115-124).  perm[i] = index (before the shuffle) of the element that ends at position i.  The
 * permutation is whatever the C++ standard library this library is built with produces -- the same one a reference
 * binary built with the same toolchain uses. */
int elemdp_epoch_permutation(int32_t n, int32_t seed, int32_t* perm);
/* name of the dominant kernel (for matching rocprofv3 rows) */
const char* elemdp_kernel_name(void);

#ifdef __cplusplus
}
#endif
#endif /* ELEMDP_H */
