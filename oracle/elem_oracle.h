/* elem_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and only as the
 * checker / reported baseline.  The product (rnaelem_amd/, libelemdp.so) never links or calls it.
 *
 * The oracle is a plain single-file C++ restatement of the reference's CPU algorithm for the
 * inside/outside/CYK hot path (see elem_oracle.cpp for the file:line map).  Parity status:
 * PINNED -- validated against the reference's own known-answer tests (PATH_COUNT, EMISSION_COUNT,
 * BPP_RNAFOLD) and against outputs of the compiled reference (oracle/_ref) committed under
 * tests/golden/.
 */
#ifndef ELEM_ORACLE_H
#define ELEM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
  ORC_NO_RSS = 1 << 0,        /* --no-rss      (motif_model.hpp:43)  */
  ORC_NO_PRF = 1 << 1,        /* --no-profile  (motif_model.hpp:44)  */
  ORC_NO_ENE = 1 << 2,        /* --no-energy   (energy_model.hpp:28) */
  ORC_THETA_SOFTMAX = 1 << 3, /* --theta-softmax */
  ORC_LIK_RATIO = 1 << 4,     /* --lik-ratio   (motif_trainer.hpp:156-202) */
  /* the reference's compile-time debug switches (const_options.hpp:12-24), runtime here */
  ORC_DBG_NO_THETA = 1 << 8,
  ORC_DBG_FIX_RSS = 1 << 9,
  ORC_DBG_NO_TURN = 1 << 10,
};

typedef struct orc_model orc_model;

/* par_text: ViennaRNA-2.0-format parameter text (e.g. rnaelem_amd/data/turner2004.elempar) */
orc_model* orc_create(const char* pattern, const char* par_text, int max_span, int max_iloop,
                      double min_bpp, double tau, int flags);
void orc_destroy(orc_model*);
const char* orc_last_error(void);

int orc_n_param(orc_model*);   /* sum of theta row widths + 2 */
int orc_n_state(orc_model*);   /* S */
int orc_n_node(orc_model*);    /* M */
/* pack_params order (motif_model.hpp:147-157); theta rows (or s rows when softmax) then lambda */
void orc_get_params(orc_model*, double* x);
void orc_set_params(orc_model*, const double* x);

/* JSON dump of the pattern automaton (same keys as `ref_dump hmm`) */
int orc_hmm_json(orc_model*, char* buf, int cap);
/* copy of one energy table by name (stack, hairpin, ... see elem_oracle.cpp) ; returns count */
int orc_energy_table(orc_model*, const char* name, double* out, int cap);
double orc_hairpin_energy(orc_model*, const uint8_t* seq, int L, int i, int j);
double orc_loop_energy(orc_model*, const uint8_t* seq, int L, int i, int j, int p, int q);
double orc_sum_ext_m(orc_model*, const uint8_t* seq, int L, int i, int j, int ext);

/* K1: plain McCaskill + BPP filter for one sequence (energy_model.hpp:188-276).
 * lnbpp / kept are (L+1)*(W+1) arrays indexed [i*(W+1)+d]; lnbpp computed on the canonical mask. */
int orc_bpp(orc_model*, const uint8_t* seq, int L, double* lnbpp, uint8_t* kept, double* bpp_eff,
            double* lnZ);

typedef struct {
  double Zo, Zari, Znasi; /* part_func(true,true), (true,false), (false,true) */
  double f;               /* Zo - Zx */
  double bpp_eff;
  int skipped;            /* non-finite Z => sequence skipped (motif_trainer.hpp:211-215) */
  int L, W;
} orc_seq_result;

/* One sequence through the training schedule of motif_trainer.hpp:204-227 (--no-shuffle, normal
 * mode).  qual has L+1 entries (char-33).  fix_rss may be NULL.  Optional outputs (may be NULL):
 * ENo/ENx [n_param-2] flattened theta-row order, EHo/EHx [2], inside_o [(L+1)*S],
 * inside/outside tables [(L+1)*(W+1)*7*S] (outside = after the FIRST, full-terminal pass). */
int orc_train_seq(orc_model*, const uint8_t* seq, int L, const uint8_t* qual, const char* fix_rss,
                  orc_seq_result* res, double* ENo, double* EHo, double* ENx, double* EHx,
                  double* inside_o, double* inside, double* outside, double* outside_o);

/* Whole-batch fn / gr evaluation == RNAelemTrainer::operator() (motif_trainer.hpp:595-633) with
 * --no-shuffle, batch = everything.  Sequences concatenated; off[n_seq+1]; qual_off likewise. */
int orc_train_eval(orc_model*, const double* x, const uint8_t* seqs, const int32_t* off,
                   const uint8_t* quals, const int32_t* qoff, int n_seq, int n_threads, double* fn,
                   double* gr, double* sum_eff, int32_t* n_skipped);

typedef struct {
  int Ys, Ye;
  double exist_prob; /* exp(logsumexp(PysL)) */
  double ZL, ZeL, PyNL;
} orc_scan_result;

/* One sequence through RNAelemScanDP (motif_scanner.hpp:204-252).  start[L], end[L+1], inner[L]
 * (log), psihat[L] (int), rss[L] chars, EN[n_param-2] (+= accumulated). */
int orc_scan_seq(orc_model*, const uint8_t* seq, int L, const uint8_t* qual, orc_scan_result* res,
                 double* start, double* end, double* inner, int32_t* psihat, char* rss, double* EN);

#ifdef __cplusplus
}
#endif
#endif
