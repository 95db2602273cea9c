// motif_trainer_gpu.hpp -- the RNAelem host bound to libelemdp: what INTEGRATION.md tells a maintainer of the reference to add.
//
// TEST INFRASTRUCTURE (oracle/): compiled only where /root/reference exists (`make -C oracle ref_gpu`), against the reference's
// own headers, into oracle/_ref/RNAelem_gpu.  Nothing in rnaelem_amd/ or in bench.py's timed region uses it.  Its purpose: the
// reference's UNCHANGED optimizer (Lbfgsb, the embedded L-BFGS-B 2.1 of RNAelem/optimizer.hpp:262-334 / :342-2788), bounds,
// regulariser, model writer and log lines drive the GPU evaluation through the C ABI, so that `RNAelem_gpu train --no-shuffle`
// can be compared with `RNAelem train --no-shuffle` iteration by iteration (tests/test_round3_gpu.py).
//
// RNAelemTrainerGpu replaces exactly one thing of RNAelemTrainer (RNAelem/motif_trainer.hpp:461-634): the evaluation
// `int operator()(V const& x, double& fn, V& gr)` (:595-633), whose work -- FASTQ records -> RNAelemTrainDP per sequence
// (:124-272) -> sums -- becomes elemdp_load_batch (once) + elemdp_train_eval (per call).  Scope: the full-batch `--no-shuffle`
// mode (TR_NO_SHUFFLE without TR_MASK / TR_ARRAY); the other modes are served by `python -m rnaelem_amd.cli`.
#pragma once
#include <cstdint>
#include <fstream>
#include <sstream>
#include <vector>

#include "application.hpp"     // the reference's headers (-I$(REF)/RNAelem)
#include "motif_trainer.hpp"

#include "elemdp.h"            // this repository's C ABI (-Iinclude)

namespace iyak {

class RNAelemTrainerGpu : public RNAelemTrainer {
  elemdp_handle* h_ = nullptr;
  int n_rec_ = 0;

  static void fail(const char* what) { die(string(what) + ": " + elemdp_last_error()); }

 public:
  // model options as the reference's command line gave them (application.hpp), batch = every record of app.seq_fname
  RNAelemTrainerGpu(App const& app, int device = 0) : RNAelemTrainer(app.tr_mode, 1) {
    check((app.tr_mode & TR_NO_SHUFFLE) and not (app.tr_mode & (TR_MASK | TR_ARRAY | TR_ARRAYEVAL)),
          "RNAelem_gpu: only the full-batch --no-shuffle mode is bound here");
    string par_text;
    elemdp_model_desc d{};
    d.pattern = app.pattern.c_str();
    if ("~T2004~" == app.ene_param_fname or "~A2007~" == app.ene_param_fname) d.energy_param = app.ene_param_fname.c_str();
    else {                                    // a parameter file: the library takes its text
      std::ifstream f(app.ene_param_fname);
      check(!!f, "cannot read energy parameter file");
      std::stringstream ss; ss << f.rdbuf(); par_text = ss.str();
      d.energy_param = par_text.c_str();
    }
    d.max_span = app.max_span; d.max_iloop = app.max_iloop; d.min_bpp = app.min_bpp; d.tau = app.tau;
    d.flags = (app.no_rss ? ELEMDP_NO_RSS : 0) | (app.no_prf ? ELEMDP_NO_PROFILE : 0) | (app.no_ene ? ELEMDP_NO_ENERGY : 0) |
              (app.theta_softmax ? ELEMDP_THETA_SOFTMAX : 0) | ((app.tr_mode & TR_LIK_RATIO) ? ELEMDP_LIK_RATIO : 0);
    d.device = device;
    if (elemdp_create(&d, &h_)) fail("elemdp_create");
    // the records as FastqReader yields them (fastq_io.hpp:64-108): base codes 0..4, qualities char - 33, L + 1 of them
    std::vector<uint8_t> codes, qual;
    std::vector<int32_t> so{0}, qo{0};
    FastqReader qr;
    qr.set_fq_fname(app.seq_fname);
    while (not qr.is_end()) {
      string id, rss; VI seq, q;
      qr.get_read(id, seq, q, rss);
      for (int c : seq) codes.push_back((uint8_t)c);
      for (int v : q) qual.push_back((uint8_t)v);
      so.push_back((int32_t)codes.size()); qo.push_back((int32_t)qual.size());
    }
    n_rec_ = (int)so.size() - 1;
    check(0 < n_rec_, "no record in the fastq file");
    if (elemdp_load_batch(h_, codes.data(), so.data(), qual.data(), qo.data(), nullptr, n_rec_)) fail("elemdp_load_batch");
  }
  ~RNAelemTrainerGpu() { if (h_) elemdp_destroy(h_); }
  RNAelemTrainerGpu(RNAelemTrainerGpu const&) = delete;

  // same contract as RNAelemTrainer::operator() (motif_trainer.hpp:595-633): unregularised fn and gr of the whole batch at x;
  // the counters and log lines the optimizer loop and the tests read are kept
  int operator()(V const& x, double& fn, V& gr) {
    _motif->unpack_params(x);
    gr.assign(size(x), 0.);
    double eff = 0.; int32_t skipped = 0;
    if (elemdp_train_eval(h_, x.data(), (int32_t)size(x), &fn, gr.data(), &eff, &skipped)) fail("elemdp_train_eval");
    _sum_eff = eff;
    if (0 == _opt.fdfcount()) cry("considered BP:", _sum_eff / n_rec_);
    ++_cnt;
    cry("iter:", _adam.itercount(), ", y:", fn, ", |gr|:", norm2(gr), ", p|x|:", _adam.rgl_term(x));
    return 0;
  }

  // RNAelemTrainer::train (motif_trainer.hpp:563-593) with the minimiser bound to THIS class's evaluation (the base class's
  // operator() is not virtual): initial parameters, bounds and regulariser are the reference's own
  void train_gpu(RNAelem& model) {
    _motif = &model;
    _motif->set_lambda(_lambda_init);
    _motif->pack_params(_params);
    check((int)size(_params) == elemdp_n_param(h_), "parameter vector of the model and of the library differ in length");
    set_bounds(model);
    set_regularization(model);
    lap();
    _cnt = 0;
    _opt.minimize(_params, *this);
    _motif->unpack_params(_opt.best_x());
    if (_motif->theta_softmax()) _motif->mm.calc_theta();
    cry("wall clock time per eval:", lap() / _cnt);
  }
};

}  // namespace iyak
