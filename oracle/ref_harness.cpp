// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A small driver of our own around the *reference's* headers, compiled by oracle/Makefile from
// the sources where they lie under /root/reference (nothing is copied into this repo; the binary
// goes to oracle/_ref/ which is git-ignored).  It is used
//   (1) in the build container to generate the golden vectors under tests/golden/
//       (tests/golden/make_golden.py is the committed generating script), and
//   (2) on the GPU box as the timed CPU baseline of kind "reference" (`time` sub-command).
//
// Sub-commands print one JSON document on stdout (doubles with 17 significant digits):
//   hmm <pattern>                         ProfileHMM tables        (profile_hmm.hpp:206-463)
//   energy [par-file]                     EnergyParam tables       (energy_param.hpp:61-84)
//   bpp <fq> <W> <C> <min_bpp>            lnBPP + kept pairs       (energy_model.hpp:188-266)
//   eval <model> <fq> [threads]           fn / gr / sum_eff        (motif_trainer.hpp:595-633)
//   evalx <fq> <pattern> [k=v ...]        same, model built from options like main.cpp:89-101
//   dp <fq> (<model>|pattern=.. k=v ...)  per-sequence Z's, inside_o, EN/EH of both outside passes
//   time <fq> <threads> <reps> (<model>|pattern=.. k=v ...)   seconds per eval
//   pathcount <pattern> <seq> <rss>       (REF_DBG build only) exp(Z), exp(Zout), ENo
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#define private public  /* dump-only access to EnergyParam's tables */
#include "util.hpp"
#include "energy_param.hpp"
#undef private
#include "motif_model.hpp"
#include "motif_trainer.hpp"
#include "motif_scanner.hpp"
#include "motif_io.hpp"

using namespace iyak;

static void pd(double x) {
  if (x == -inf) printf("\"-inf\"");
  else if (x == inf) printf("\"inf\"");
  else if (x != x) printf("\"nan\"");
  else printf("%.17g", x);
}
static void pv(V const& v) {
  printf("[");
  for (size_t i = 0; i < v.size(); ++i) { if (i) printf(","); pd(v[i]); }
  printf("]");
}
static void pvv(VV const& v) {
  printf("[");
  for (size_t i = 0; i < v.size(); ++i) { if (i) printf(","); pv(v[i]); }
  printf("]");
}
static void parr(const char* name, double const* a, int n, bool last = false) {
  printf("\"%s\":[", name);
  for (int i = 0; i < n; ++i) { if (i) printf(","); pd(a[i]); }
  printf("]%s\n", last ? "" : ",");
}
static void pis(std::vector<IS> const& v) {
  printf("[");
  for (size_t i = 0; i < v.size(); ++i) printf("%s%d", i ? "," : "", v[i].id);
  printf("]");
}

struct Opts {
  std::map<string, string> kv;
  string get(string k, string d) const { auto it = kv.find(k); return it == kv.end() ? d : it->second; }
  double getd(string k, double d) const { auto it = kv.find(k); return it == kv.end() ? d : atof(it->second.c_str()); }
  int geti(string k, int d) const { auto it = kv.find(k); return it == kv.end() ? d : atoi(it->second.c_str()); }
};
static Opts parse_opts(int argc, char** argv, int from) {
  Opts o;
  for (int i = from; i < argc; ++i) {
    string a(argv[i]);
    size_t p = a.find('=');
    if (p == string::npos) o.kv["model"] = a; else o.kv[a.substr(0, p)] = a.substr(p + 1);
  }
  return o;
}

/* build a model either from a model file or from options (defaults = application.hpp:76-300) */
static void build_model(RNAelem& model, Opts const& o) {
  if (o.kv.count("model")) {
    RNAelemReader reader;
    reader.set_model_fname(o.get("model", ""));
    reader.read_model(model);
  } else {
    model.set_theta_softmax(o.geti("theta_softmax", 0));
    model.set_hyper_param(o.getd("rho_s", 0.1), o.getd("rho_theta", 0.1), o.getd("rho_lambda", 0.1),
                          o.getd("tau", 0.1), o.getd("lambda_prior", 0));
    model.set_energy_params(o.get("ene", "~T2004~"), o.geti("max_span", 50), o.geti("max_iloop", 30),
                            o.getd("min_bpp", 1e-4), o.geti("no_ene", 0));
    model.set_motif_pattern(o.get("pattern", "(.....)"), o.geti("no_rss", 0), o.geti("no_prf", 0));
    model.set_lambda(o.getd("lambda", 0.));
  }
  if (o.kv.count("lambda0")) model._lambda[0] = o.getd("lambda0", 0);
  if (o.kv.count("lambda1")) model._lambda[1] = o.getd("lambda1", 0);
  if (o.kv.count("x")) { /* explicit parameter vector, pack_params order */
    V x = split<double>(o.get("x", ""), ",");
    V cur; model.pack_params(cur);
    check(size(x) == size(cur), "bad x length", size(x), size(cur));
    model.unpack_params(x);
  }
  if (o.kv.count("xseed")) { /* deterministic pseudo-random perturbation of theta / lambda */
    V cur; model.pack_params(cur);
    unsigned long long s = (unsigned long long)o.geti("xseed", 1) * 0x9E3779B97F4A7C15ull + 12345;
    for (size_t i = 0; i < cur.size(); ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      double u = double(s >> 11) / 9007199254740992.0; /* [0,1) */
      if (i + 2 >= cur.size()) cur[i] = 0.2 + 1.5 * u; /* lambda > 0 */
      else cur[i] += (u - 0.5) * 2.0;
    }
    model.unpack_params(cur);
  }
}

#ifndef REF_DBG
static int cmd_hmm(string pattern) {
  ProfileHMM mm;
  mm.build(pattern);
  int M = (int)mm.size();
  int S = (int)mm.state().size();
  printf("{\"pattern\":\"%s\",\"reg_pattern\":\"%s\",\"M\":%d,\"S\":%d,\n", pattern.c_str(),
         mm.reg_pattern().c_str(), M, S);
  printf("\"node\":\"");
  for (int h = 0; h < M; ++h) printf("%c", (char)mm.node(h));
  printf("\",\n\"theta_id\":[");
  for (int h = 0; h < M; ++h) printf("%s%d", h ? "," : "", mm.theta_id(h));
  printf("],\n\"theta_sizes\":[");
  for (size_t i = 0; i < mm.theta().size(); ++i) printf("%s%d", i ? "," : "", (int)mm.theta()[i].size());
  printf("],\n\"theta\":"); pvv(mm.theta());
  printf(",\n\"state\":[");
  for (int s = 0; s < S; ++s) printf("%s[%d,%d]", s ? "," : "", mm.state()[s].l, mm.state()[s].r);
  printf("],\n\"loop_state\":"); pis(mm.loop_state());
  printf(",\n\"reachable\":[");
  for (int a = 0; a < M; ++a) {
    printf("%s[", a ? "," : "");
    for (int b = 0; b < M; ++b) printf("%s%d", b ? "," : "", (int)mm.reachable(a, b));
    printf("]");
  }
  printf("],\n\"right\":[");
  for (int s = 0; s < S; ++s) { if (s) printf(","); pis(mm.loop_right_trans(s)); }
  printf("],\n\"left\":[");
  for (int s = 0; s < S; ++s) { if (s) printf(","); pis(mm.loop_left_trans(s)); }
  printf("],\n\"pair\":[");
  for (int s = 0; s < S; ++s) { if (s) printf(","); pis(mm.pair_trans(s)); }
  printf("],\n\"loop_loop\":[");
  bool first = true;
  for (auto const& q : mm.loop_loop_states()) {
    printf("%s[%d,%d,%d,%d]", first ? "" : ",", q[0].id, q[1].id, q[2].id, q[3].id);
    first = false;
  }
  printf("]}\n");
  return 0;
}

static int cmd_energy(string fname) {
  EnergyModel em;
  em.set_param_file(fname);
  EnergyParam& ep = em.ep();
  printf("{\n");
  parr("stack", &ep._stack[0][0], 49);
  parr("hairpin", ep._hairpin, 31);
  parr("bulge", ep._bulge, 31);
  parr("internal", ep._internal, 31);
  parr("ninio", ep._ninio, 31);
  parr("mismatch_h", &ep._mismatch_h[0][0][0], 175);
  parr("mismatch_i", &ep._mismatch_i[0][0][0], 175);
  parr("mismatch_m", &ep._mismatch_m[0][0][0], 175);
  parr("mismatch_1ni", &ep._mismatch_1ni[0][0][0], 175);
  parr("mismatch_23i", &ep._mismatch_23i[0][0][0], 175);
  parr("mismatch_ext", &ep._mismatch_ext[0][0][0], 175);
  parr("dangle5", &ep._dangle5[0][0], 40);
  parr("dangle3", &ep._dangle3[0][0], 40);
  parr("int_11", &ep._int_11[0][0][0][0], 8 * 8 * 5 * 5);
  parr("int_21", &ep._int_21[0][0][0][0][0], 8 * 8 * 5 * 5 * 5);
  /* int_22: only entries with all four unpaired bases in 1..4 and types 1..6 are read by the
     reference's parser (energy_param.hpp:604-608); the rest is partly uninitialised memory */
  printf("\"int_22_acgu\":[");
  bool first = true;
  for (int a = 1; a < 7; ++a) for (int b = 1; b < 7; ++b)
    for (int c = 1; c < 5; ++c) for (int d = 1; d < 5; ++d)
      for (int e = 1; e < 5; ++e) for (int f = 1; f < 5; ++f) {
        if (!first) printf(",");
        pd(ep._int_22[a][b][c][d][e][f]);
        first = false;
      }
  printf("],\n");
  parr("triloop", ep._triloop, 40);
  parr("tetraloop", ep._tetraloop, 40);
  parr("hexaloop", ep._hexaloop, 40);
  printf("\"triloops\":\"%s\",\"tetraloops\":\"%s\",\"hexaloops\":\"%s\",\n", ep._triloops.c_str(),
         ep._tetraloops.c_str(), ep._hexaloops.c_str());
  printf("\"term_au\":"); pd(ep._term_au);
  printf(",\"mlintern\":"); pd(ep._mlintern);
  printf(",\"mlclosing\":"); pd(ep._mlclosing);
  printf(",\"ml_base\":"); pd(ep._ml_base);
  printf(",\"lxc37\":"); pd(ep._lxc37);
  printf("}\n");
  return 0;
}

static int cmd_bpp(string fq, int W, int C, double min_bpp) {
  RNAelem model;
  model.set_theta_softmax(false);
  model.set_energy_params("~T2004~", W, C, min_bpp, false);
  model.set_hyper_param(0.1, 0.1, 0.1, 0.1, -1);
  FastqReader qr;
  qr.set_fq_fname(fq);
  printf("{\"W\":%d,\"C\":%d,\"min_bpp\":%.17g,\"seqs\":[\n", W, C, min_bpp);
  bool first = true;
  while (not qr.is_end()) {
    string id, rss; VI seq, qual;
    qr.get_read(id, seq, qual, rss);
    int L = size(seq);
    model.set_seq(seq); /* runs the BPP filter when min_bpp>0 */
    printf("%s{\"id\":\"%s\",\"L\":%d,\"bpp_eff\":", first ? "" : ",\n", id.c_str(), L);
    pd(model.em.bpp_eff());
    /* kept pairs after the filter */
    printf(",\"kept\":[");
    bool f2 = true;
    for (int i = 0; i <= L; ++i) for (int j = i + 1; j <= std::min(L, i + model.em.max_pair()); ++j)
      if (model.em.is_parsable<EM::ST_P>(i, j)) { printf("%s[%d,%d]", f2 ? "" : ",", i, j); f2 = false; }
    printf("]");
    /* ln BPP of every candidate (recomputed on the unfiltered canonical mask) */
    {
      RNAelem m2;
      m2.set_theta_softmax(false);
      m2.set_energy_params("~T2004~", W, C, 0., false);
      m2.set_hyper_param(0.1, 0.1, 0.1, 0.1, -1);
      m2.set_seq(seq);
      m2.em.calc_BPP();
      printf(",\"lnZ\":"); pd(m2.em.inside_o(L));
      printf(",\"lnbpp\":[");
      bool f3 = true;
      for (int i = 0; i <= L; ++i) for (int j = i + 1; j <= std::min(L, i + m2.em.max_pair()); ++j)
        if (m2.em.is_parsable<EM::ST_P>(i, j)) {
          printf("%s[%d,%d,", f3 ? "" : ",", i, j); pd(m2.em.lnBPP(i, j)); printf("]");
          f3 = false;
        }
      printf("]");
    }
    printf("}");
    first = false;
  }
  printf("]}\n");
  return 0;
}

static int cmd_eval(string fq, Opts const& o) {
  RNAelem model;
  build_model(model, o);
  RNAelemTrainer t(TR_NORMAL | TR_NO_SHUFFLE | (o.geti("lik", 0) ? TR_LIK_RATIO : 0), o.geti("threads", 1));   // lik=1: --lik-ratio
  t.set_fq_name(fq);
  t.set_conditions(1, 1e-5, 0, 2, -1, "~NULL~");
  V x; model.pack_params(x);
  t.eval(model);
  printf("{\"n_seq\":%d,\"x\":", t._qr.N()); pv(x);
  printf(",\n\"fn\":"); pd(t._fn);
  printf(",\n\"gr\":"); pv(t._gr);
  printf(",\n\"sum_eff\":"); pd(t._sum_eff);
  printf("}\n");
  return 0;
}

/* per-sequence dump following the schedule of motif_trainer.hpp:204-227 by hand */
class DumpDP : public RNAelemTrainDP {
 public:
  using RNAelemTrainDP::RNAelemTrainDP;
  void run(string const& fq, bool full) {
    FastqReader qr;
    qr.set_fq_fname(fq);
    printf("{\"S\":%d,\"M\":%d,\"seqs\":[\n", _m.S, _m.M);
    bool first = true;
    while (not qr.is_end()) {
      VI qual;
      qr.get_read(_id, _seq, qual, _rss);
      _m.set_seq(_seq);
      _m.set_ws(qual);
      init_inside_tables();
      init_outside_tables(true, true);
      _m.compute_inside(InsideFun(this, ws()));
      double Zo = part_func(true, true), Za = part_func(true, false), Zn = part_func(false, true);
      VV ENo, ENx; V EHo{0., 0.}, EHx{0., 0.};
      _m.mm.clear_emit_count(ENo);
      _m.mm.clear_emit_count(ENx);
      printf("%s{\"id\":\"%s\",\"L\":%d,\"W\":%d,\"positive\":%d,\"bpp_eff\":", first ? "" : ",\n",
             _id.c_str(), _m.L, _m.W, int(!(-inf < ws().back())));
      pd(_m.no_rss() ? 0. : _m.em.bpp_eff());
      printf(",\"Zo\":"); pd(Zo); printf(",\"Zari\":"); pd(Za); printf(",\"Znasi\":"); pd(Zn);
      printf(",\"inside_o\":"); pvv(_inside_o);
      if (std::isfinite(Zo) and std::isfinite(Za)) {
        _m.compute_outside(OutsideFun(this, ws(), Zo, EHo, ENo));
        printf(",\"outside_o_full\":"); pvv(_outside_o);
        double insum = 0, outsum = 0; long nfin = 0, nfout = 0;
        if (!_m.no_rss()) for (auto& a : _inside) for (auto& b : a) for (auto& c : b) for (auto d : c)
          if (d > -inf) { insum += d; ++nfin; }
        if (!_m.no_rss()) for (auto& a : _outside) for (auto& b : a) for (auto& c : b) for (auto d : c)
          if (d > -inf) { outsum += d; ++nfout; }
        printf(",\"inside_finite\":%ld,\"inside_sum\":", nfin); pd(insum);
        printf(",\"outside_finite\":%ld,\"outside_sum\":", nfout); pd(outsum);
        if (full and !_m.no_rss()) {
          printf(",\"inside\":[");
          bool f = true;
          for (int i = 0; i <= _m.L; ++i) for (int d = 0; d <= _m.W; ++d) for (int e = 0; e < _m.E - 1; ++e)
            for (int s = 0; s < _m.S; ++s) if (_inside[i][d][e][s] > -inf) {
              printf("%s[%d,%d,%d,%d,", f ? "" : ",", i, d, e, s); pd(_inside[i][d][e][s]); printf("]");
              f = false;
            }
          printf("],\"outside\":[");
          f = true;
          for (int i = 0; i <= _m.L; ++i) for (int d = 0; d <= _m.W; ++d) for (int e = 0; e < _m.E - 1; ++e)
            for (int s = 0; s < _m.S; ++s) if (_outside[i][d][e][s] > -inf) {
              printf("%s[%d,%d,%d,%d,", f ? "" : ",", i, d, e, s); pd(_outside[i][d][e][s]); printf("]");
              f = false;
            }
          printf("]");
        }
        double Zx;
        if (-inf < ws().back()) { init_outside_tables(false, true); Zx = Zn; }
        else { init_outside_tables(true, false); Zx = Za; }
        _m.compute_outside(OutsideFun(this, ws(), Zx, EHx, ENx));
        printf(",\"f\":"); pd(Zo - Zx);
        printf(",\"ENo\":"); pvv(ENo); printf(",\"EHo\":"); pv(EHo);
        printf(",\"ENx\":"); pvv(ENx); printf(",\"EHx\":"); pv(EHx);
      }
      printf("}");
      first = false;
    }
    printf("]}\n");
  }
};

static int cmd_dp(string fq, Opts const& o) {
  RNAelem model;
  build_model(model, o);
  RNAelemTrainer t(TR_NORMAL | TR_NO_SHUFFLE, 1);
  DumpDP f(model, t._from, t._to, t._sum_eff, t._mx_input, t._mx_update, t._qr, t._mode, 0, 2);
  f.run(fq, o.geti("full", 0));
  return 0;
}

static int cmd_time(string fq, int threads, int reps, Opts const& o) {
  RNAelem model;
  build_model(model, o);
  RNAelemTrainer t(TR_NORMAL | TR_NO_SHUFFLE, threads);
  t.set_fq_name(fq);
  t.set_conditions(1, 1e-5, 0, 2, -1, "~NULL~");
  t._motif = &model;
  t.set_bounds(model);
  V x; model.pack_params(x);
  double fn; V gr;
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) t(x, fn, gr);
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("{\"n_seq\":%d,\"threads\":%d,\"reps\":%d,\"sec_per_eval\":%.6f,\"seq_per_sec\":%.6f,\"fn\":",
         t._qr.N(), threads, reps, sec / reps, t._qr.N() * reps / sec);
  pd(fn);
  printf("}\n");
  return 0;
}
#else /* REF_DBG: the reference's test configuration (RNAelem-test/test.cpp:74-86) */
class DbgDP : public RNAelemTrainDP {
 public:
  using RNAelemTrainDP::RNAelemTrainDP;
};
static int cmd_pathcount(string pattern, string seq, string rss) {
  RNAelem model;
  RNAelemTrainer t(TR_NORMAL | TR_NO_SHUFFLE, 1);
  model.set_theta_softmax(false);
  model.set_energy_params("~T2004~", large, large, 0., true);
  model.set_hyper_param(0., 0., 0., 1., -1.);
  t.set_conditions(-1, 1e-4, 0, 2, -1, "~NULL~");
  DbgDP f(model, t._from, t._to, t._sum_eff, t._mx_input, t._mx_update, t._qr, t._mode, 0, 2);
  f._m.set_motif_pattern(pattern);
  seq_str2int(seq, f._seq);
  f._rss = rss;
  f._m.em.fix_rss(rss);
  f._m.set_seq(f._seq);
  f._m.set_ws(VI(size(seq) + 1, 1));
  V EHo{0., 0.}; VV ENo;
  f._m.mm.clear_emit_count(ENo);
  f.init_inside_tables();
  f.init_outside_tables();
  f._m.compute_inside(RNAelemTrainDP::InsideFun(&f, f.ws()));
  f._m.compute_outside(RNAelemTrainDP::OutsideFun(&f, f.ws(), oneL, EHo, ENo));
  printf("{\"pattern\":\"%s\",\"seq\":\"%s\",\"rss\":\"%s\",\"Z\":", pattern.c_str(), seq.c_str(), rss.c_str());
  pd(expL(f.part_func())); printf(",\"Zout\":"); pd(expL(f.part_func_outside()));
  printf(",\"ENo\":"); pvv(ENo); printf("}\n");
  return 0;
}
#endif

int main(int argc, char** argv) {
  init_ostream(4);
  if (argc < 2) { fprintf(stderr, "usage: see header comment\n"); return 2; }
  string c(argv[1]);
  try {
#ifndef REF_DBG
    if (c == "hmm" && argc >= 3) return cmd_hmm(argv[2]);
    if (c == "energy") return cmd_energy(argc >= 3 ? argv[2] : "~T2004~");
    if (c == "bpp" && argc >= 6) return cmd_bpp(argv[2], atoi(argv[3]), atoi(argv[4]), atof(argv[5]));
    if (c == "eval" && argc >= 4) {
      Opts o = parse_opts(argc, argv, 4); o.kv["model"] = argv[2];
      return cmd_eval(argv[3], o);
    }
    if (c == "evalx" && argc >= 4) {
      Opts o = parse_opts(argc, argv, 4); o.kv["pattern"] = argv[3];
      return cmd_eval(argv[2], o);
    }
    if (c == "dp" && argc >= 4) return cmd_dp(argv[2], parse_opts(argc, argv, 3));
    if (c == "shuffle" && argc >= 5) {   // shuffle <seq> <k> <iter_cnt>: the negative of motif_trainer.hpp:145-152
      string sq(argv[2]);
      srand((int)std::count(sq.begin(), sq.end(), sq[0]) + atoi(argv[4]));
      ushuffle::set_randfunc(long_rand);
      std::vector<char> neg(sq.size() + 1, 0);
      ushuffle::shuffle(sq.c_str(), neg.data(), (int)sq.size(), atoi(argv[3]));
      printf("%s\n", neg.data());
      return 0;
    }
    if (c == "time" && argc >= 6) return cmd_time(argv[2], atoi(argv[3]), atoi(argv[4]), parse_opts(argc, argv, 5));
#else
    if (c == "pathcount" && argc >= 5) return cmd_pathcount(argv[2], argv[3], argv[4]);
#endif
  } catch (std::runtime_error& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  fprintf(stderr, "bad command\n");
  return 2;
}
