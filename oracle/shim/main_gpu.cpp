// RNAelem_gpu -- `RNAelem [train] --no-shuffle ...` with the evaluation on the GPU (oracle/shim/motif_trainer_gpu.hpp).
// TEST INFRASTRUCTURE: the reference's option parser, model set-up, optimizer, model writer and (CPU) scanner, unchanged, around
// libelemdp.  Same command lines as the reference binary: `train` (RNAelem/main.cpp:85-115) and no sub-command = train, write
// the model to --out1, scan the training set to --out2 (main.cpp:47-84, what script/elem spawns).  Device = $ELEMDP_DEVICE.
#include <cstdlib>
#include <iostream>

#include "const_options.hpp"
#include "util.hpp"
#include "profile_hmm.hpp"
#include "motif_model.hpp"
#include "motif_trainer.hpp"
#include "motif_scanner.hpp"
#include "motif_io.hpp"
#include "application.hpp"

#include "motif_trainer_gpu.hpp"

using namespace iyak;

int main(int const argc, char const* argv[]) {
  try {
    App app(argc, argv);
    check(App::PM_TRAIN == app.mode or App::PM_NORMAL == app.mode, "RNAelem_gpu: only training is bound (`train`, or no sub-command)");
    RNAelem model;
    if ("~NONE~" != app.model_fname) {
      RNAelemReader reader;
      reader.set_model_fname(app.model_fname);
      reader.read_model(model);
    } else {
      model.set_theta_softmax(app.theta_softmax);
      model.set_hyper_param(app.rho_s, app.rho_theta, app.rho_lambda, app.tau, app.lambda_prior);
      model.set_energy_params(app.ene_param_fname, app.max_span, app.max_iloop, app.min_bpp, app.no_ene);
      model.set_motif_pattern(app.pattern, app.no_rss, app.no_prf);
    }
    const char* dev = std::getenv("ELEMDP_DEVICE");
    RNAelemTrainerGpu train(app, dev ? std::atoi(dev) : 0);
    train.set_conditions(app.max_iter, app.eps, app.lambda_init, app.kmer_shuf, app.batch_size, app.out1);
    train.train_gpu(model);
    RNAelemWriter writer;
    if (App::PM_NORMAL == app.mode) writer.set_out_id(1);
    writer.write(model);
    if (App::PM_NORMAL == app.mode) {      // the reference's own scanner (CPU) on the model the GPU trained
      RNAelemScanner scan(app.thread);
      scan.set_out_id(2);
      scan.set_fq_name(app.seq_fname);
      scan.scan(model);
    }
  } catch (std::exception& e) {
    std::cerr << e.what() << std::endl;
    return 1;
  }
  return 0;
}
