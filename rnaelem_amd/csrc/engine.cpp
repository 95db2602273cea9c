// engine.cpp -- host side of libelemdp.so: model set-up, batch residency, kernel launches, C ABI.
//
// Mirrors the part of the reference that sits directly around the hot path:
//   RNAelem::set_motif_pattern / set_energy_params      RNAelem/motif_model.hpp:72-97
//   RNAelem::pack_params / unpack_params                 RNAelem/motif_model.hpp:147-168
//   RNAelemTrainer::operator()  (fn / gr assembly)       RNAelem/motif_trainer.hpp:248-271, 595-633
//   RNAelemScanner::scan                                 RNAelem/motif_scanner.hpp:938-949
// There is deliberately no CPU implementation of the DP in this library.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <numeric>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include "../../include/elemdp.h"
#include "automaton.h"
#include "device_layout.h"
#include "energy_tables.h"
#include "host_prep.h"
#include "kernels.h"
#include "lin_params.h"

namespace elemdp {
namespace {

thread_local std::string g_error;
std::string g_data_dir;

struct HipError : std::runtime_error {
  explicit HipError(const std::string& m) : std::runtime_error(m) {}
};
struct ArgError : std::runtime_error {
  explicit ArgError(const std::string& m) : std::runtime_error(m) {}
};
struct StateError : std::runtime_error {
  explicit StateError(const std::string& m) : std::runtime_error(m) {}
};

#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
  } while (0)

// Makes `device` the calling thread's current HIP device for the lifetime of the guard and restores the previous one.  Every
// computing entry point of an Engine takes one: the current device is a per-thread setting, so a handle used from another
// host thread (the prefetch thread of the mini-batch loop), or a second handle on another GPU, must not inherit whatever
// device was current.
class DeviceGuard {
 public:
  explicit DeviceGuard(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    if (prev_ != device) { HIP_OK(hipSetDevice(device)); changed_ = true; }
  }
  ~DeviceGuard() { if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;

 private:
  int prev_ = -1;
  bool changed_ = false;
};

// RCCL entry points, resolved at run time: the library has no link-time dependency on librccl (hosts that bring their own
// collective -- torch.distributed in rnaelem_amd/distributed.py -- never load it)
struct Rccl {
  typedef struct { char internal[128]; } UniqueId;           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
  typedef void* Comm;                                         // ncclComm_t
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  static constexpr int kDouble = 8, kSum = 0;                 // ncclFloat64, ncclSum
  static Rccl& get() {
    static Rccl r;
    static std::once_flag once;   // (engines are driven from two host threads when a batch is streamed)
    std::call_once(once, [] {
      void* lib = nullptr;
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
      if (lib) {
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(lib, "ncclAllReduce"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
      }
    });
    return r;
  }
  bool ok() const { return GetUniqueId && CommInitRank && AllReduce && CommDestroy; }
};
#define RCCL_OK(expr)                                                                                                     \
  do {                                                                                                                    \
    int r_ = (expr);                                                                                                      \
    if (r_ != 0)                                                                                                          \
      throw elemdp::HipError(std::string(#expr) + ": " +                                                                  \
                             (elemdp::Rccl::get().GetErrorString ? elemdp::Rccl::get().GetErrorString(r_) : "rccl error")); \
  } while (0)

// owning device buffer
class DevBuf {
 public:
  DevBuf() = default;
  ~DevBuf() { reset(); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  // (an allocation that is large enough and at most twice too large is kept: fresh device memory costs ~20 ms / GB, and
  // the plan sets of a load are rebuilt chunk after chunk with similar sizes; bytes() is the capacity)
  // slack: allocate an eighth more than asked for -- buffers whose size depends on the data (interior-loop items of a batch):
  // the next batch of the same shape then fits without a re-allocation
  // (such a buffer is also never given up for a smaller one: the last chunk of a load is smaller than the others)
  void alloc(size_t bytes, bool slack = false) {
    bytes = bytes ? bytes : 8;
    if (p_ && bytes <= bytes_ && (slack || bytes_ <= 2 * bytes + (size_t(1) << 20))) return;
    reset();
    bytes_ = slack ? bytes + bytes / 8 : bytes;
    HIP_OK(hipMalloc(&p_, bytes_));
  }
  void reset() { if (p_) { (void)hipFree(p_); p_ = nullptr; bytes_ = 0; } }
  template <class T> T* as() const { return static_cast<T*>(p_); }
  size_t bytes() const { return bytes_; }
  template <class T> void upload(const std::vector<T>& v, hipStream_t st) {
    alloc(v.size() * sizeof(T));
    if (!v.empty()) HIP_OK(hipMemcpyAsync(p_, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
  }

 private:
  void* p_ = nullptr;
  size_t bytes_ = 0;
};

// pinned host staging buffer, kept from load to load (a fresh std::vector of 24 MB costs its page faults -- 30 ms per
// 10 000 x L=300 for the four arrays of load_batch -- and a pageable upload goes through the runtime's own staging copies)
class HostBuf {
 public:
  HostBuf() = default;
  ~HostBuf() { if (p_) (void)hipHostFree(p_); }
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  template <class T> T* get(size_t n) {
    const size_t bytes = std::max<size_t>(n * sizeof(T), 8);
    if (bytes > bytes_) {
      if (p_) (void)hipHostFree(p_);
      p_ = nullptr;
      bytes_ = bytes + bytes / 8;
      HIP_OK(hipHostMalloc(&p_, bytes_, hipHostMallocDefault));
    }
    return static_cast<T*>(p_);
  }

 private:
  void* p_ = nullptr;
  size_t bytes_ = 0;
};

std::string default_data_dir() {
  if (!g_data_dir.empty()) return g_data_dir;
  if (const char* e = std::getenv("ELEMDP_DATA_DIR")) return e;
  Dl_info info;
  if (dladdr(reinterpret_cast<void*>(&default_data_dir), &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t k = p.find_last_of('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/data";
  }
  return "data";
}

std::string read_file(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw ArgError("cannot open energy parameter file: " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

// one set of plan arrays (for a subset of the batch)
struct PlanSet {
  int first = 0, count = 0;
  std::vector<SeqPlan> h;
  DevBuf d_plans, dmin, e_stack, e_ext, e_ml, e_close, e_hp, off_outer, off_inner, off_left, off_right, cursor, items,
      item_in, idx_inner, idx_left, idx_right, items_inner, items_left, items_right;
  bool permuted = false;   // keep item copies in the secondary orders (resident plan of the train pipeline)
  bool inner_only = false; // build only the by_inner order (the BPP filter needs no outside values of loop cells)
  bool sorted = true;      // the segments of the role lists are sorted by item index (Engine::ensure_sorted_plan)
  PlanKernelArgs ka;       // the arguments the set was built with (for the sort, should a later evaluation want it)
  int64_t n_items = 0;
  PlanArrays arrays() const {
    PlanArrays a;
    a.dmin = dmin.as<int16_t>();
    a.e_stack = e_stack.as<double>(); a.e_ext = e_ext.as<double>(); a.e_ml = e_ml.as<double>();
    a.e_close = e_close.as<double>(); a.e_hp = e_hp.as<double>();
    a.by_outer_off = off_outer.as<int32_t>(); a.by_inner_off = off_inner.as<int32_t>();
    a.by_left_off = off_left.as<int32_t>(); a.by_right_off = off_right.as<int32_t>();
    a.cursor = cursor.as<int32_t>();
    a.items = items.as<LoopItem>(); a.item_in = item_in.as<uint8_t>();
    a.by_inner_idx = idx_inner.as<int32_t>(); a.by_left_idx = idx_left.as<int32_t>(); a.by_right_idx = idx_right.as<int32_t>();
    a.items_inner = items_inner.as<LoopItem>(); a.items_left = items_left.as<LoopItem>(); a.items_right = items_right.as<LoopItem>();
    return a;
  }
};

}  // namespace

class Engine {
 public:
  explicit Engine(const elemdp_model_desc& d);
  ~Engine();

  int n_param() const { return au_.n_theta() + 2; }
  int n_state() const { return au_.S(); }
  int n_node() const { return au_.M(); }
  const Automaton& automaton() const { return au_; }
  const AutomatonLayout& layout() const { return lays_.shadow >= 0 ? lays_ : lay_; }   // (what a train evaluation sweeps)
  bool softmax() const { return flags_ & ELEMDP_THETA_SOFTMAX; }

  void load_batch(const uint8_t* seq, const int32_t* off, const uint8_t* qual, const int32_t* qoff, const char* fix, int n);
  // reduce: all-reduce the vector over the communicator of elemdp_comm_init (if any) before it is handed out
  void train_partial(const double* x, int n_param, void* partial, bool device_ptr, bool reduce = false);
  void train_finish(const double* reduced, double* fn, double* gr, double* sum_eff, int32_t* n_skipped);
  void scan(const double* x, int n_param, elemdp_scan_out* out);
  int partial_len() const { return 4 + 2 * au_.n_theta() + 4; }
  void set_option(const std::string& key, double v);
  void comm_init(int rank, int world, const void* id);
  void comm_destroy();
  bool has_comm() const { return comm_ != nullptr; }

  int n_seq() const { return n_seq_; }
  const std::vector<SeqPlan>& plans() const { return h_plans_; }
  void seq_stats(double* out, int n);
  bool bpp_eff_known() const {
    if (!streaming_) return true;
    for (char c : st_have_eff_) if (!c) return false;
    return true;
  }
  void debug_tables(double* inside, double* outside, double* inside_o, double* outside_o, double* ENo, double* ENx, double* EH);
  void batch_pairs(int idx, uint8_t* kept, double* lnbpp, int cap);
  double last_ms[3] = {0, 0, 0};

 private:
  void upload_params(const double* x, const AutomatonLayout& lay, bool trivial);
 public:
  void set_theta_from(const double* x);  // host: theta (log-softmax of x when theta-softmax)
 private:
  void build_planset(PlanSet& ps, int first, int count, const uint32_t* d_okbits);
  void ensure_sorted_plan();
  LdsLayout lds_layout(const AutomatonLayout& lay, int Lmax, int nword_max, bool scan) const;
  DpArgs base_args(const AutomatonLayout& lay, const int32_t* d_ints, const double* d_params, const PlanSet& ps,
                   const uint32_t* d_okbits, int S);
  // row: doubles per cell of a band table slot (0: the dense layout, 7 * S; the compact tables of the scaled-linear pipeline
  // pass AutomatonLayout::tab_row)
  void ensure_slots(int S, bool scan, int n_want, int row = 0);
  void run_train(bool first_pass_only);
  void run_train_batch();
  void run_lin_batch();
  int prepare_lin(LinArgs& a, bool sched1, bool dense_too = false, int n_eval = 0);
  int balanced_group(size_t per_slot_bytes, int n = 0);
  int group_cap_ = 8192;   // most sequences swept in lockstep (a first scan uses fewer: fresh table memory costs ~20 ms / GB)
  TrArgs log_pipeline_args();
  void init_device();
  void flatten_automaton();
  void poison_tables();
  void lin_weights(int first = 0, int count = 0);
  void upload_automaton();
  bool opt_prune_ = true;   // transition lists pruned to the transitions of complete parses (Automaton::flatten)
  int opt_row_pad_ = 1;     // rows of the compact tables padded to a multiple of this many doubles (8 = 64-byte lines; 1 = none)
  bool opt_cell_major_ = false;   // compact tables cell by cell (the seven rows of a cell side by side) instead of plane by plane
  // a train evaluation covers the records [eval_first, eval_first + eval_count) of the resident batch only (count 0: all): the
  // mini-batch trainer loads the records + negatives of several coming evaluations as ONE batch -- the filter and the plan do not
  // depend on x, and a load of 128 sequences costs as much as one of 1024 (launch-bound) -- and evaluates them range by range
  int opt_eval_first_ = 0, opt_eval_count_ = 0;
  std::vector<int32_t> h_order_r_;
  DevBuf d_order_r_, d_plans_sorted_r_;
  int range_key_[2] = {-1, -1};
  bool opt_det_ = false;    // deterministic reductions of the scaled-linear train evaluation (LinArgs::det): bit-identical repeats
  bool opt_fast_ = true;    // table-driven unary phases of the train kernels (lin_fast.h); 0 = the generic rule code
  int opt_nblk_ = 0;        // blocks of cells per band-kernel workgroup (LinArgs::nblk): 0 = chosen per launch, n = n wherever they fit
  bool opt_poison_ = false; // tests: the table slots are filled with NaN before every evaluation of the scaled-linear pipeline, so
                            // that a read of an entry nobody stored shows up in the results (the compact tables hold garbage there)
  void require_device() const;
  bool has_device_ = false;
  int want_device_ = -1;
  Rccl::Comm comm_ = nullptr;   // in-library collective (elemdp_comm_init): all-reduce of the partial vector in train_eval
  int comm_rank_ = 0, comm_world_ = 1;
  // ---- streaming: a batch that is not kept resident as a whole (option "max_resident", or more sequences than the device
  // memory holds).  The handle then keeps the host copy of the input and two inner engines on the same device: chunk k is
  // evaluated (or scanned) on one while the other runs load_batch (BPP filter + plan) of chunk k+1 on a second host thread;
  // the partial vectors of the chunks are summed in chunk order.  The reference streams its records the same way
  // (motif_trainer.hpp:124-153 over fastq_io.hpp:132-167) and recomputes the BPP filter in every evaluation, too.
  bool streaming_ = false;
  int st_chunk_ = 0, opt_max_resident_ = 0;
  // an inner handle of a streamed batch: never streams itself, and its table slots stay inside the share of the device
  // memory the outer handle left for them (two inner handles: plan + slots of each within 2/5 of the free memory)
  bool inner_ = false;
  size_t slot_budget_ = 0, st_slot_budget_ = 0;
  std::vector<uint8_t> st_seq_, st_qual_;
  std::vector<char> st_fix_;
  std::vector<double> st_rows_;                             // [Z(ari,nasi), Z(ari), Z(nasi), f, skipped] per sequence
  std::unique_ptr<Engine> sub_[2];
  std::vector<std::pair<std::string, double>> opt_log_;     // options to replay on the inner engines
  elemdp_model_desc desc_;
  std::string desc_pattern_, desc_par_;
  bool desc_has_par_ = false;
  bool should_stream(const int32_t* off, int n);
  void stream_setup(const uint8_t* seq, const int32_t* off, const uint8_t* qual, const int32_t* qoff, const char* fix, int n);
  void stream_load_chunk(int k, Engine& e);
  // The BPP filter does not depend on the parameters: a streamed batch keeps the filtered pair mask (1.3 KB per sequence of
  // L = 200) and the kept fractions of every chunk on the host after its first load; later loads of the chunk hand them
  // to the inner handle, which then skips K1 and only rebuilds the plan.
  std::vector<std::vector<uint32_t>> st_mask_;
  std::vector<std::vector<double>> st_eff_;
  std::vector<char> st_have_eff_;        // chunk k has been loaded at least once: its bpp_eff values are known
  const uint32_t* preset_bits_ = nullptr;   // (consumed by the next load_batch)
  const double* preset_eff_ = nullptr;
  size_t preset_words_ = 0;
  size_t bits_words_ = 0;                   // words of the pair mask of the loaded batch
  void set_filter_preset(const uint32_t* bits, size_t words, const double* eff) { preset_bits_ = bits; preset_words_ = words; preset_eff_ = eff; }
  void filter_result(std::vector<uint32_t>& bits, std::vector<double>& eff) {
    bits.resize(bits_words_);
    HIP_OK(hipMemcpyAsync(bits.data(), d_okbits1_.as<void>(), sizeof(uint32_t) * bits_words_, hipMemcpyDeviceToHost, st_));
    HIP_OK(hipStreamSynchronize(st_));
    eff.resize(h_plans_.size());
    for (size_t k = 0; k < h_plans_.size(); ++k) eff[k] = h_plans_[k].bpp_eff;
  }
  void stream_subs();
  template <class Work> void stream_chunks(Work work);
  void stream_train(const double* x, int n_param, void* partial, bool device_ptr, bool reduce);
  void stream_scan(const double* x, int n_param, elemdp_scan_out* out);

  Automaton au_;
  EnergyTables et_;
  AutomatonLayout lay_, lay0_, layr_;
  std::vector<int32_t> ints_, ints0_, intsr_;
  bool linear_ok_ = true;
  int flags_, max_span_, max_iloop_;
  bool loops_finite_ = false;   // loop_tables_finite(et_)
  bool bpp_planes_finite_ = false;   // the linear filter's planes hold nothing but finite values (cleared at allocation)
  double min_bpp_, tau_;
  int device_ = -1, n_cu_ = 256;   // device_ < 0: no HIP device (host-only handle)
  hipStream_t st_ = nullptr;
  hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev2_[2] = {nullptr, nullptr};
  bool opt_two_streams_ = true;
  // groups evaluated concurrently (run_lin_batch, scan): stream k of gs_ (gs_[0] = st_) with its second-pass stream gs2_[k]
  static constexpr int kMaxGroupStreams = 4;
  hipStream_t gs_[kMaxGroupStreams] = {nullptr, nullptr, nullptr, nullptr};
  // (created when first used: the runtime deals its few hardware queues to streams in the order they are made, and two
  // handles that evaluate and load at the same time -- the mini-batch trainer -- must not end up sharing one)
  void need_group_streams(int ns) {
    for (int k = 1; k < ns && k < kMaxGroupStreams; ++k)
      if (!gs_[k]) HIP_OK(hipStreamCreateWithFlags(&gs_[k], hipStreamNonBlocking));
  }
  hipEvent_t gev_[kMaxGroupStreams][2] = {}, gdone_[kMaxGroupStreams] = {}, gstart_ = nullptr;
  int opt_group_streams_ = 2;
  DevBuf d_et_, d_xet_, d_ints_, d_ints0_, d_params_, d_params0_, d_counter_, d_lay_, d_lay0_, d_layr_, d_intsr_;
  std::vector<double> theta_;  // log-probabilities of the last evaluation (softmax Jacobian)

  // batch
  int n_seq_ = 0, Lmax_ = 0, Wmax_ = 0, nword_max_ = 0;
  std::vector<SeqPlan> h_plans_;
  std::vector<int32_t> h_order_, h_seq_off_, h_qual_off_;
  DevBuf d_seq_, d_ws_, d_unp_, d_ndot_, d_okbits0_, d_okbits1_, d_order_, d_ncanon_, d_zero_ws_;
  std::vector<double> h_lnbpp_;          // optional (keep_lnbpp)
  std::vector<int64_t> h_lnbpp_base_;
  PlanSet plan_;
  // slots
  int n_slots_ = 0, slots_S_ = 0, slot_override_ = 0;
  bool slots_scan_ = false;
  size_t band_stride_ = 0, ext_stride_ = 0;
  DevBuf d_band_in_, d_band_out_, d_ext_in_, d_ext_out_, d_tr_ext_, d_tr_stack_, d_tmp_;
  DevBuf d_seq_out_, d_partial_;
  int out_stride_ = 0;
  // options
  int opt_slots_ = 0;
  bool opt_keep_lnbpp_ = false;
  bool opt_first_pass_only_ = false;
  bool opt_profile_ = false;
  // 4 = scaled-linear batch pipeline (lin_kernels.hip), 3 = log-space batch pipeline, 2 = fused one-workgroup-per-sequence kernel
  int opt_pipeline_ = 4;
  bool opt_sorted_plan_ = false;   // option "sorted_plan": sort the role lists at load_batch whatever the pipeline
  // scaled-linear pipeline
  AutomatonLayout lays_;                 // the automaton with the shadow copy of (0,0): both outside passes in one sweep
  std::vector<int32_t> intss_;
  DevBuf d_lays_, d_intss_;
  int tables_S_ = 0;                     // state stride of the tables of the last linear evaluation (debug_tables)
  std::vector<double> last_x_;           // parameters of the last train evaluation (debug_tables repeats it with the generic kernels)
  std::vector<double> h_lin_, h_lins_;   // linear parameter block of the last evaluation (plain automaton / with the shadow state)
  std::vector<uint8_t> h_seq_;           // base codes of the batch (table export)
  HostBuf hb_ws_, hb_ews_, hb_unp_;      // staging of load_batch
  int64_t n_cells_total_ = 0;
  bool tables_linear_ = false;           // the resident tables hold scaled linear values (debug_tables converts)
  int n_flagged_last_ = 0;
  DevBuf d_a_in_, d_a_out_;   // pair tables of the factorised rule 2 (lin_rules.h), per slot [W+1][Lmax+1][n_ap]
  DevBuf d_plans_sorted_;   // plan records in processing (h_order_) order
  DevBuf d_ews_, d_xwc_, d_xwi_, d_lin_, d_lins_, d_zs_, d_flagged_, d_det_;
  int lin_slots_ = 0;
  int opt_schedule_ = 1;   // 1 = linear (ari pass + one-state nasi pass), 0 = the reference's two full passes
  int opt_dbg_ = 0;        // timing experiments (LinArgs::dbg); results are wrong when set
  int opt_group_ = 0;      // sequences swept in lockstep by the batch pipeline (0 = auto)
  DevBuf d_prof_;
  DevBuf d_bpp_band_in_, d_bpp_band_out_, d_bpp_ext_in_, d_bpp_ext_out_, d_bpp_tmp_;   // S = 1 tables of the BPP filter
  PlanSet bpp_plan_;   // plan over the unfiltered mask, chunk by chunk (only the filter reads it)
  DevBuf d_bpp_order_, d_bpp_rows_, d_bpp_kept_, d_okbits_end_, d_nitems_, d_plans_all_;   // scratch kept across loads
  DevBuf d_bpp_plans_, d_bpp_xw_, d_bpp_dmin_, d_bpp_cand_, d_bpp_plist_, d_bpp_poff_;   // linear-semiring filter (bpp_kernels.hip)
  bool opt_bpp_log_ = false;                      // option "bpp_log": the log-space filter over the unfiltered plan
 public:
  std::vector<long long> last_prof;
 private:
};

Engine::Engine(const elemdp_model_desc& d)
    : au_(d.pattern ? d.pattern : ""), flags_(d.flags), max_span_(d.max_span), max_iloop_(d.max_iloop), min_bpp_(d.min_bpp),
      tau_(d.tau) {
  desc_ = d;
  desc_pattern_ = d.pattern ? d.pattern : "";
  desc_has_par_ = d.energy_param != nullptr;
  if (desc_has_par_) desc_par_ = d.energy_param;
  if ((flags_ & ELEMDP_NO_RSS) && (flags_ & ELEMDP_NO_PROFILE)) throw ArgError("no-rss, no-profile are exclusive.");
  if ((flags_ & ELEMDP_NO_RSS) && au_.reg_pattern().find(')') != std::string::npos)
    throw ArgError("search pattern must not include pair when no-rss mode");
  if (max_span_ < 1) throw ArgError("max_span must be positive");
  if (au_.S() > 128) throw ArgError("pattern has more than 128 interval states (not supported by this build)");
  if (!(tau_ > 0)) throw ArgError("tau must be positive");
  std::string par = d.energy_param ? d.energy_param : "~T2004~";
  if (par == "~T2004~") par = read_file(default_data_dir() + "/turner2004.elempar");
  else if (par == "~A2007~") par = read_file(default_data_dir() + "/andronescu2007.elempar");
  parse_energy_text(par, &et_);
  if (const char* e = getenv("ELEMDP_POISON")) opt_poison_ = e[0] == '1';   // (the test-suite sets it: see poison_tables)
  flatten_trivial(&lay0_, &ints0_);
  // the linear schedule needs state 0 = (0,0) to be closed under every transition family (decided before the automaton is
  // flattened: the shadow copy of (0,0) is only built for a closed state)
  linear_ok_ = au_.state(0).l == 0 && au_.state(0).r == 0;
  for (int c : au_.right(0)) linear_ok_ = linear_ok_ && c == 0;
  for (int c : au_.left(0)) linear_ok_ = linear_ok_ && c == 0;
  for (int c : au_.pair(0)) linear_ok_ = linear_ok_ && c == 0;
  for (auto const& sp : au_.splits(0)) linear_ok_ = linear_ok_ && sp[0] == 0 && sp[1] == 0;
  for (auto const& q : au_.quads()) if (q[0] == 0) linear_ok_ = linear_ok_ && q[1] == 0 && q[2] == 0 && q[3] == 0;
  flatten_automaton();

  int ndev = 0;
  has_device_ = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
  want_device_ = d.device;
  // Without a GPU the handle still serves the host-only calls (describe, initial_params,
  // train_finish); everything that computes raises ELEMDP_ENODEV -- there is no CPU path.
  if (has_device_) init_device();
}

// the flat transition lists of the pattern automaton (pruned to the transitions that can occur in a complete parse unless
// option "prune" = 0) and of its restriction to state (0,0)
void Engine::flatten_automaton() {
  au_.flatten(&lay_, &ints_, false, opt_prune_, false, opt_row_pad_, opt_cell_major_);
  au_.flatten(&layr_, &intsr_, true, opt_prune_, false, opt_row_pad_, opt_cell_major_);
  if (linear_ok_ && au_.S() < 127) au_.flatten(&lays_, &intss_, false, opt_prune_, true, opt_row_pad_, opt_cell_major_);
  else { lays_ = lay_; lays_.shadow = -1; intss_ = ints_; }
  lin_slots_ = 0;   // (the pair tables of the linear pipeline are sized by the automaton's pair list)
}

void Engine::upload_automaton() {
  d_ints_.upload(ints_, st_);
  d_intsr_.upload(intsr_, st_);
  d_lay_.alloc(sizeof(AutomatonLayout));
  d_layr_.alloc(sizeof(AutomatonLayout));
  d_lays_.alloc(sizeof(AutomatonLayout));
  d_intss_.upload(intss_, st_);
  HIP_OK(hipMemcpyAsync(d_lay_.as<void>(), &lay_, sizeof(AutomatonLayout), hipMemcpyHostToDevice, st_));
  HIP_OK(hipMemcpyAsync(d_layr_.as<void>(), &layr_, sizeof(AutomatonLayout), hipMemcpyHostToDevice, st_));
  HIP_OK(hipMemcpyAsync(d_lays_.as<void>(), &lays_, sizeof(AutomatonLayout), hipMemcpyHostToDevice, st_));
  HIP_OK(hipStreamSynchronize(st_));
}

void Engine::require_device() const {
  if (!has_device_) throw HipError("no HIP device available (libelemdp has no CPU path)");
}

void Engine::init_device() {
  if (want_device_ >= 0) { device_ = want_device_; HIP_OK(hipSetDevice(device_)); }
  else HIP_OK(hipGetDevice(&device_));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, device_));
  n_cu_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_OK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
  gs_[0] = st_;
  for (int k = 0; k < kMaxGroupStreams; ++k) {
    if (k) for (auto& e : gev_[k]) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&gdone_[k], hipEventDisableTiming));
  }
  HIP_OK(hipEventCreateWithFlags(&gstart_, hipEventDisableTiming));
  for (auto& e : ev_) HIP_OK(hipEventCreate(&e));
  for (auto& e : ev2_) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  gev_[0][0] = ev2_[0]; gev_[0][1] = ev2_[1];
  d_et_.alloc(sizeof(EnergyTables));
  HIP_OK(hipMemcpyAsync(d_et_.as<void>(), &et_, sizeof(EnergyTables), hipMemcpyHostToDevice, st_));
  loops_finite_ = loop_tables_finite(et_);
  {   // Boltzmann weights of the loop tables, for the BPP filter (energy_rules.h: loop_weight)
    std::unique_ptr<EnergyTables> x(new EnergyTables);
    exp_tables(et_, x.get());
    d_xet_.alloc(sizeof(EnergyTables));
    HIP_OK(hipMemcpy(d_xet_.as<void>(), x.get(), sizeof(EnergyTables), hipMemcpyHostToDevice));
    std::unique_ptr<BppCandTable> ct(new BppCandTable);
    build_bpp_cand(*x, ct.get());
    d_bpp_cand_.alloc(sizeof(BppCandTable));
    HIP_OK(hipMemcpy(d_bpp_cand_.as<void>(), ct.get(), sizeof(BppCandTable), hipMemcpyHostToDevice));
  }
  d_ints0_.upload(ints0_, st_);
  upload_automaton();
  d_lay0_.alloc(sizeof(AutomatonLayout));
  d_lin_.alloc(sizeof(double) * (kLinEth + au_.n_theta() + 1));
  HIP_OK(hipMemcpyAsync(d_lay0_.as<void>(), &lay0_, sizeof(AutomatonLayout), hipMemcpyHostToDevice, st_));
  d_params_.alloc(sizeof(ParamBlock) + sizeof(double) * (au_.n_theta() + 1));
  d_params0_.alloc(sizeof(ParamBlock) + sizeof(double));
  d_counter_.alloc(sizeof(int32_t));
  d_partial_.alloc(sizeof(double) * partial_len());
  {  // parameters of the one-state automaton of the BPP filter: lambda = 1, no emissions
    ParamBlock pb;
    pb.lambda[0] = pb.lambda[1] = 1.;
    pb.log_tau = 0.;
    pb.lam_same = 1;
    pb.pad = 0;
    HIP_OK(hipMemcpyAsync(d_params0_.as<void>(), &pb, sizeof(pb), hipMemcpyHostToDevice, st_));
  }
  HIP_OK(hipStreamSynchronize(st_));
}

void Engine::comm_init(int rank, int world, const void* id) {
  require_device();
  if (!id || world < 1 || rank < 0 || rank >= world) throw ArgError("elemdp_comm_init: bad rank / world / id");
  Rccl& r = Rccl::get();
  if (!r.ok()) throw HipError("no HIP device available for the collective: librccl.so could not be loaded");
  DeviceGuard dg(device_);
  comm_destroy();
  Rccl::UniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  RCCL_OK(r.CommInitRank(&comm_, world, uid, rank));
  comm_rank_ = rank;
  comm_world_ = world;
}

void Engine::comm_destroy() {
  if (comm_) { DeviceGuard dg(device_); (void)Rccl::get().CommDestroy(comm_); comm_ = nullptr; comm_world_ = 1; comm_rank_ = 0; }
}

Engine::~Engine() {
  if (!has_device_) return;
  DeviceGuard dg(device_);
  comm_destroy();
  if (st_) (void)hipStreamSynchronize(st_);
  for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
  for (auto& e : ev2_) if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < kMaxGroupStreams; ++k) {
    if (k) for (auto& e : gev_[k]) if (e) (void)hipEventDestroy(e);
    if (gdone_[k]) (void)hipEventDestroy(gdone_[k]);
    if (k && gs_[k]) (void)hipStreamDestroy(gs_[k]);
  }
  if (gstart_) (void)hipEventDestroy(gstart_);
  if (st_) (void)hipStreamDestroy(st_);
}

void Engine::set_option(const std::string& key, double v) {
  if (key == "max_resident") { opt_max_resident_ = (int)v; return; }
  if (key == "slots") opt_slots_ = (int)v;
  else if (key == "keep_lnbpp") opt_keep_lnbpp_ = v != 0;
  else if (key == "first_pass_only") opt_first_pass_only_ = v != 0;
  else if (key == "profile") opt_profile_ = v != 0;
  else if (key == "two_streams") opt_two_streams_ = v != 0;
  else if (key == "group_streams") opt_group_streams_ = (int)v;
  else if (key == "pipeline") {
    if ((int)v != 3 && (int)v != 4) throw ArgError("option pipeline: 4 (scaled linear, default) or 3 (log space); the fused kernel (2) is retired");
    opt_pipeline_ = (int)v;
  }
  else if (key == "group") opt_group_ = (int)v;
  else if (key == "schedule") opt_schedule_ = (int)v;
  else if (key == "dbg") opt_dbg_ = (int)v;
  else if (key == "bpp_log") opt_bpp_log_ = v != 0;
  else if (key == "poison") opt_poison_ = v != 0;
  else if (key == "fast") opt_fast_ = v != 0;
  else if (key == "nblk") opt_nblk_ = std::max(0, std::min(64, (int)v));
  else if (key == "deterministic") opt_det_ = v != 0;
  else if (key == "sorted_plan") opt_sorted_plan_ = v != 0;
  else if (key == "eval_first") opt_eval_first_ = (int)v;
  else if (key == "eval_count") opt_eval_count_ = (int)v;
  else if (key == "prune" || key == "row_pad" || key == "cell_major") {
    if (key == "prune") opt_prune_ = v != 0;
    else if (key == "cell_major") opt_cell_major_ = v != 0;
    else opt_row_pad_ = std::max(1, (int)v);
    n_slots_ = 0;
    flatten_automaton();
    if (has_device_) { DeviceGuard dg(device_); HIP_OK(hipStreamSynchronize(st_)); upload_automaton(); }
  }
  else throw ArgError("unknown option: " + key);
  for (auto& m : st_mask_) m.clear();                  // (an option may change the filter: cached masks of a streamed batch go)
  opt_log_.emplace_back(key, v);                       // (replayed on the inner engines of a streamed batch)
  for (auto& e : sub_) if (e) e->set_option(key, v);
}

void Engine::set_theta_from(const double* x) {
  const int nt = au_.n_theta();
  theta_.assign(x, x + nt);
  if (softmax()) {  // theta = log-softmax of the score rows (ProfileHMM::calc_theta, profile_hmm.hpp:103-111)
    for (int r = 0; r < au_.n_rows(); ++r) {
      const int o = au_.row_offset(r), w = au_.row_width(r);
      double tot = -std::numeric_limits<double>::infinity();
      for (int c = 0; c < w; ++c) {  // logsumexp by sequential log1p(exp()) like util.hpp:195-209
        const double a = tot, b = x[o + c];
        tot = (b == -INFINITY) ? a : (a == -INFINITY) ? b : (a < b ? b + std::log1p(std::exp(a - b)) : a + std::log1p(std::exp(b - a)));
      }
      for (int c = 0; c < w; ++c) theta_[o + c] = x[o + c] - tot;
    }
  }
}

void Engine::upload_params(const double* x, const AutomatonLayout& lay, bool) {
  const int nt = au_.n_theta();
  set_theta_from(x);
  std::vector<double> blob(sizeof(ParamBlock) / sizeof(double) + nt + 1, 0.);
  ParamBlock pb;
  pb.lambda[0] = x[nt];
  pb.lambda[1] = x[nt + 1];
  pb.log_tau = std::log(tau_);
  pb.lam_same = (x[nt] == x[nt + 1]) ? 1 : 0;
  pb.pad = 0;
  std::memcpy(blob.data(), &pb, sizeof(pb));
  std::copy(theta_.begin(), theta_.end(), blob.begin() + sizeof(ParamBlock) / sizeof(double));
  HIP_OK(hipMemcpyAsync(d_params_.as<void>(), blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice, st_));
  // linear parameter block (+ the weight tables of the table-driven unary phases) for the plain automaton and for the one
  // with the shadow state (their transition ids differ)
  make_lin_params(lay_, ints_.data(), theta_.data(), tau_, (flags_ & ELEMDP_NO_PROFILE) != 0, &h_lin_);
  d_lin_.alloc(sizeof(double) * (h_lin_.size() + 1));
  HIP_OK(hipMemcpyAsync(d_lin_.as<void>(), h_lin_.data(), h_lin_.size() * sizeof(double), hipMemcpyHostToDevice, st_));
  make_lin_params(lays_, intss_.data(), theta_.data(), tau_, (flags_ & ELEMDP_NO_PROFILE) != 0, &h_lins_);
  d_lins_.alloc(sizeof(double) * (h_lins_.size() + 1));
  HIP_OK(hipMemcpyAsync(d_lins_.as<void>(), h_lins_.data(), h_lins_.size() * sizeof(double), hipMemcpyHostToDevice, st_));
  HIP_OK(hipStreamSynchronize(st_));  // blob is a local
  (void)lay;
}

LdsLayout Engine::lds_layout(const AutomatonLayout& lay, int Lmax, int nword_max, bool scan) const {
  LdsLayout l;
  int o = 0;
  auto take = [&](int bytes) { int p = o; o += (bytes + 15) & ~15; return p; };
  l.theta = take(8 * (lay.n_theta + 1));
  l.en_o = take(8 * (lay.n_theta + 1));
  l.en_x = take(8 * (lay.n_theta + 1));
  l.eh = take(8 * 4);
  l.zs = take(8 * 8);
  l.ws = take(8 * (Lmax + 1));
  l.post = take(scan ? 8 * 3 * (Lmax + 1) : 16);
  l.ints = take(4 * lay.n_small);
  l.wave_scr = take(8 * 128 * (kThreads / 64));
  l.okbits = take(4 * nword_max);
  l.dmin = take(2 * (Lmax + 1));
  l.seq = take(Lmax + 1);
  l.unp = take(Lmax + 1);
  l.total = o;
  if (l.total > 160 * 1024) throw ArgError("sequence / pattern too large for the LDS staging of this build");
  return l;
}

// wall-clock laps on stderr when ELEMDP_TIME is set (where does a first call spend its time?)
static void dbg_lap(const char* what) {
  static const bool on = getenv("ELEMDP_TIME") != nullptr;
  static auto t0 = std::chrono::steady_clock::now();
  if (!on) return;
  const auto t1 = std::chrono::steady_clock::now();
  fprintf(stderr, "[elemdp %8.1f ms] %s\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), what);
  t0 = t1;
}

void Engine::build_planset(PlanSet& ps, int first, int count, const uint32_t* d_okbits) {
  ps.first = first;
  ps.count = count;
  ps.h.assign(h_plans_.begin() + first, h_plans_.begin() + first + count);
  int64_t dmin_b = 0, cell_b = 0, off_b = 0, bits_end = 0, ncell_max = 0;
  int lmax = 0, wmax1 = 0;
  for (auto& p : ps.h) {
    const int64_t nc = (int64_t)(p.L + 1) * (p.W + 1);
    p.dmin_base = dmin_b; p.cell_base = cell_b; p.off_base = off_b; p.item_base = 0; p.n_items = 0;
    dmin_b += p.L + 1; cell_b += nc; off_b += nc + 1;
    bits_end = std::max<int64_t>(bits_end, p.bits_base + (nc + 31) / 32);
    ncell_max = std::max(ncell_max, nc);
    lmax = std::max(lmax, (int)p.L);
    wmax1 = std::max(wmax1, (int)p.W + 1);
  }
  DevBuf& d_okbits_end = d_okbits_end_;   // the pair mask by (end, span): scratch of the item enumeration (kept across loads)
  d_okbits_end.alloc(sizeof(uint32_t) * (size_t)bits_end);
  ps.d_plans.upload(ps.h, st_);
  const bool chunked = ps.inner_only;   // the plan of the unfiltered mask is rebuilt chunk after chunk: its buffers only grow
  ps.dmin.alloc(sizeof(int16_t) * dmin_b, chunked);
  for (DevBuf* b : {&ps.e_stack, &ps.e_ext, &ps.e_ml, &ps.e_close, &ps.e_hp}) b->alloc(sizeof(double) * cell_b, chunked);
  for (DevBuf* b : {&ps.off_outer, &ps.off_inner, &ps.off_left, &ps.off_right, &ps.cursor}) b->alloc(sizeof(int32_t) * off_b, chunked);
  DevBuf& d_nitems = d_nitems_;
  d_nitems.alloc(sizeof(int32_t) * count);
  PlanKernelArgs a;
  a.et = d_et_.as<EnergyTables>();
  a.b.seq = d_seq_.as<uint8_t>(); a.b.ws = d_ws_.as<double>(); a.b.unp = d_unp_.as<uint8_t>();
  a.b.ndot = (flags_ & ELEMDP_DBG_FIX_RSS) ? d_ndot_.as<int32_t>() : nullptr;
  a.okbits = d_okbits;
  a.okbits_end = d_okbits_end.as<uint32_t>();
  a.ncell_max = (int32_t)ncell_max;
  a.n_roles = ps.inner_only ? 1 : 3;
  a.lmax = lmax;
  a.wmax1 = wmax1;
  a.nword_max = (int32_t)((ncell_max + 31) / 32);
  a.plans = ps.d_plans.as<SeqPlan>();
  a.first = 0; a.count = count;
  a.p = ps.arrays();
  a.no_ene = (flags_ & ELEMDP_NO_ENERGY) ? 1 : 0;
  a.min_span = (flags_ & ELEMDP_DBG_NO_TURN) ? 1 : 5;
  a.fix_rss = (flags_ & ELEMDP_DBG_FIX_RSS) ? 1 : 0;
  a.count_fast = (loops_finite_ && !a.fix_rss && !getenv("ELEMDP_PLAN_ENUM_COUNT")) ? 1 : 0;
  HIP_OK(launch_plan_cells(a, d_nitems.as<int32_t>(), st_));
  std::vector<int32_t> n_items(count);
  HIP_OK(hipMemcpyAsync(n_items.data(), d_nitems.as<void>(), sizeof(int32_t) * count, hipMemcpyDeviceToHost, st_));
  HIP_OK(hipStreamSynchronize(st_));
  int64_t ib = 0;
  for (int k = 0; k < count; ++k) {
    ps.h[k].n_items = n_items[k]; ps.h[k].item_base = ib; ib += n_items[k];
    a.nitems_max = std::max(a.nitems_max, n_items[k]);
  }
  ps.n_items = ib;
  if (getenv("ELEMDP_PLAN_DEBUG")) fprintf(stderr, "planset: %d sequences, %lld items, largest %d, cells %lld\n", count, (long long)ib, a.nitems_max, (long long)ncell_max);
  ps.d_plans.upload(ps.h, st_);
  ps.items.alloc(sizeof(LoopItem) * ib, true);
  ps.item_in.alloc(ib, true);
  for (DevBuf* b : {&ps.idx_inner, &ps.idx_left, &ps.idx_right}) b->alloc(sizeof(int32_t) * ib, true);
  if (ps.permuted) for (DevBuf* b : {&ps.items_inner, &ps.items_left, &ps.items_right}) b->alloc(sizeof(LoopItem) * (ib + 1), true);
  a.plans = ps.d_plans.as<SeqPlan>();
  a.p = ps.arrays();
  // The scaled-linear pipeline adds over the role lists with atomics: their order inside a segment is immaterial, and the sort
  // is a quarter of the plan builder.  The log-space pipeline and the deterministic mode get it (here, or later through
  // ensure_sorted_plan when the option arrives after the batch).
  a.sort_roles = (ps.inner_only || opt_det_ || opt_pipeline_ != 4 || opt_sorted_plan_) ? 1 : 0;
  HIP_OK(launch_plan_items(a, st_));
  if (ps.permuted && !plan_copies_fused(a)) HIP_OK(launch_permute_items(a, st_));
  HIP_OK(hipStreamSynchronize(st_));
  ps.sorted = a.sort_roles != 0;
  ps.ka = a;
}

void Engine::ensure_sorted_plan() {
  if (plan_.sorted || plan_.count <= 0) return;
  HIP_OK(launch_plan_sort(plan_.ka, st_));
  if (plan_.permuted) HIP_OK(launch_permute_items(plan_.ka, st_));
  HIP_OK(hipStreamSynchronize(st_));
  plan_.sorted = true;
}

void Engine::ensure_slots(int S, bool scan, int n_want, int row) {
  if (row <= 0) row = kNumBandStates * S;
  const size_t band = (size_t)(Wmax_ + 1) * (Lmax_ + 1) * row;
  // (the log-space fallback of the scaled-linear pipeline sweeps dense tables over the same buffers: at least one fits)
  const size_t dense1 = (size_t)kNumBandStates * (Wmax_ + 1) * (Lmax_ + 1) * au_.S();
  const size_t ext = (size_t)(Lmax_ + 1) * S;
  int want = opt_slots_ > 0 ? opt_slots_ : 2 * n_cu_;
  if (slot_override_ > 0) want = slot_override_;
  want = std::max(1, std::min(want, n_want));
  size_t free_b = 0, total_b = 0;
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  const size_t per_slot = (band + ext) * 2 * sizeof(double) + (scan ? ext * sizeof(TraceRec) + 16 * (Lmax_ + 2) : 0);
  if (n_slots_ >= want && slots_S_ == S && band_stride_ == band && (slots_scan_ || !scan)) return;
  // (no reset: DevBuf::alloc keeps what is large enough -- a load_batch per evaluation must not re-allocate the tables)
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  const size_t held_t = d_band_in_.bytes() + d_band_out_.bytes() + d_ext_in_.bytes() + d_ext_out_.bytes() + d_tmp_.bytes();
  const size_t held_tr = d_tr_ext_.bytes() + d_tr_stack_.bytes();
  size_t budget = (size_t)((double)(free_b + held_t + held_tr) * 0.72);
  if (slot_budget_ > 0) budget = std::min(budget, slot_budget_);
  if (per_slot * (size_t)want > budget) want = (int)std::max<size_t>(1, budget / per_slot);
  if (per_slot * want > free_b + held_t + held_tr) throw HipError("not enough device memory for one table slot");
  n_slots_ = want; slots_S_ = S; slots_scan_ = scan;
  band_stride_ = band; ext_stride_ = ext;
  d_band_in_.alloc(std::max(band * want, dense1) * sizeof(double));
  d_band_out_.alloc(std::max(band * want, dense1) * sizeof(double));
  d_ext_in_.alloc(ext * want * sizeof(double));
  d_ext_out_.alloc(ext * want * sizeof(double));
  d_tmp_.alloc(ext * 3 * want * sizeof(double));
  if (scan) {
    d_tr_ext_.alloc(ext * want * sizeof(TraceRec));
    d_tr_stack_.alloc((size_t)want * 4 * (4 * (Lmax_ + 2)) * sizeof(int32_t));
  }
}

DpArgs Engine::base_args(const AutomatonLayout& lay, const int32_t* d_ints, const double* d_params, const PlanSet& ps,
                         const uint32_t* d_okbits, int S) {
  DpArgs a;
  std::memset(&a, 0, sizeof(a));
  a.lay = lay;
  a.layp = (&lay == &lay0_) ? d_lay0_.as<AutomatonLayout>() : d_lay_.as<AutomatonLayout>();
  a.ints = d_ints;
  a.params = d_params;
  a.no_prf = (flags_ & ELEMDP_NO_PROFILE) ? 1 : 0;
  a.m_min = (flags_ & ELEMDP_DBG_NO_TURN) ? 4 : 10;
  a.no_rss = (flags_ & ELEMDP_NO_RSS) ? 1 : 0;
  a.plans = ps.d_plans.as<SeqPlan>();
  a.n_seq = ps.count;
  a.counter = d_counter_.as<int32_t>();
  a.b.seq = d_seq_.as<uint8_t>(); a.b.ws = d_ws_.as<double>(); a.b.unp = d_unp_.as<uint8_t>();
  a.b.ndot = nullptr;
  a.okbits = d_okbits;
  a.p = ps.arrays();
  a.band_in = d_band_in_.as<double>(); a.band_out = d_band_out_.as<double>();
  a.ext_in = d_ext_in_.as<double>(); a.ext_out = d_ext_out_.as<double>();
  a.band_stride = (size_t)kNumBandStates * (Wmax_ + 1) * (Lmax_ + 1) * S;
  a.ext_stride = (size_t)(Lmax_ + 1) * S;
  a.tmp = d_tmp_.as<double>();
  a.tmp_stride = a.ext_stride;
  return a;
}

void Engine::load_batch(const uint8_t* seq, const int32_t* off, const uint8_t* qual, const int32_t* qoff, const char* fix,
                        int n) {
  require_device();
  DeviceGuard dg(device_);
  // a rejected batch leaves the handle without a batch (ELEMDP_ESTATE for what follows) instead of the new sizes over the old
  // device buffers
  n_seq_ = 0;
  streaming_ = false;
  if (n > 0 && seq && off && qual && qoff && should_stream(off, n)) { stream_setup(seq, off, qual, qoff, fix, n); return; }
  if (n <= 0 || !seq || !off || !qual || !qoff) throw ArgError("load_batch: empty batch or null pointer");
  const bool fixmode = flags_ & ELEMDP_DBG_FIX_RSS;
  if (fixmode && !fix) throw ArgError("load_batch: ELEMDP_DBG_FIX_RSS needs fix_rss strings");
  const bool no_rss = flags_ & ELEMDP_NO_RSS;
  h_plans_.assign(n, SeqPlan());
  h_seq_off_.assign(off, off + n + 1);
  h_qual_off_.assign(qoff, qoff + n + 1);
  int64_t seq_b = 0, pos_b = 0, bits_b = 0;
  Lmax_ = Wmax_ = nword_max_ = 0;
  for (int k = 0; k < n; ++k) {
    const int L = off[k + 1] - off[k];
    if (L <= 0) throw ArgError("load_batch: empty sequence");
    if (L > 32000) throw ArgError("load_batch: sequence longer than 32000");
    if (qoff[k + 1] - qoff[k] != L + 1) throw ArgError("bad seq format. (quality must have L+1 entries)");  // motif_trainer.hpp:139
    SeqPlan& p = h_plans_[k];
    p.L = L;
    p.W = std::min(L, max_span_);
    p.C = std::min(p.W - 2 - ((flags_ & ELEMDP_DBG_NO_TURN) ? 2 : 5), max_iloop_);  // energy_model.hpp:271-273
    p.positive = qual[qoff[k + 1] - 1] == 0;
    p.seq_base = seq_b; p.pos_base = pos_b; p.bits_base = bits_b;
    const int nword = (int)(((int64_t)(L + 1) * (p.W + 1) + 31) / 32);
    seq_b += L; pos_b += L + 1; bits_b += nword;
    Lmax_ = std::max(Lmax_, L); Wmax_ = std::max(Wmax_, p.W); nword_max_ = std::max(nword_max_, nword);
    p.bpp_eff = 0.;
  }
  if ((double)kNumBandStates * (Wmax_ + 1) * (Lmax_ + 1) * au_.S() >= 2147483648.0)
    throw ArgError("load_batch: sequence too long for this pattern (band table exceeds 2^31 entries)");
  // static arrays
  h_seq_.resize((size_t)seq_b);
  std::vector<uint8_t>& h_seq = h_seq_;
  uint8_t* h_unp = hb_unp_.get<uint8_t>((size_t)pos_b);
  double* h_ws = hb_ws_.get<double>((size_t)pos_b);
  double* h_ews = hb_ews_.get<double>((size_t)pos_b);
  std::memset(h_unp, 1, (size_t)pos_b);
  std::vector<int32_t> h_ndot;
  std::vector<uint32_t> h_bits;
  if (fixmode) { h_ndot.assign((size_t)pos_b, 0); h_bits.assign((size_t)bits_b, 0u); }
  for (int k = 0; k < n; ++k) {
    const SeqPlan& p = h_plans_[k];
    for (int t = 0; t < p.L; ++t) {
      const uint8_t c = seq[off[k] + t];
      if (c > 4) throw ArgError("load_batch: base code out of range");
      h_seq[p.seq_base + t] = c;
    }
    position_weights(qual + qoff[k], p.L + 1, &h_ws[p.pos_base], &h_ews[p.pos_base]);
    if (fixmode) {
      const char* f = fix + off[k];
      nondot_prefix(f, p.L, &h_ndot[p.pos_base]);
      std::vector<int> open;
      for (int t = 0; t < p.L; ++t) {
        h_unp[p.pos_base + t] = f[t] == '.';
        if (f[t] == '(') open.push_back(t);
        else if (f[t] == ')') {
          if (open.empty()) throw ArgError("bad rss: unbalanced");
          const int o = open.back();
          open.pop_back();
          const int d = t + 1 - o;
          if (d > p.W) throw ArgError("bad rss: pair wider than max_span");
          const int64_t c = (int64_t)o * (p.W + 1) + d;
          h_bits[p.bits_base + (c >> 5)] |= 1u << (c & 31);
        } else if (f[t] != '.') throw ArgError(std::string("bad rss char: ") + f[t]);
      }
      if (!open.empty()) throw ArgError("bad rss: unbalanced");
    }
  }
  dbg_lap("load: host arrays");
  h_order_.resize(n);
  std::iota(h_order_.begin(), h_order_.end(), 0);
  std::stable_sort(h_order_.begin(), h_order_.end(), [&](int a, int b) { return h_plans_[a].L > h_plans_[b].L; });
  d_seq_.upload(h_seq, st_);
  d_ws_.alloc(sizeof(double) * pos_b);
  HIP_OK(hipMemcpyAsync(d_ws_.as<void>(), h_ws, sizeof(double) * pos_b, hipMemcpyHostToDevice, st_));
  d_ews_.alloc(sizeof(double) * pos_b);
  HIP_OK(hipMemcpyAsync(d_ews_.as<void>(), h_ews, sizeof(double) * pos_b, hipMemcpyHostToDevice, st_));
  d_zero_ws_.alloc(sizeof(double) * pos_b);
  HIP_OK(hipMemsetAsync(d_zero_ws_.as<void>(), 0, sizeof(double) * pos_b, st_));
  d_unp_.alloc((size_t)pos_b);
  HIP_OK(hipMemcpyAsync(d_unp_.as<void>(), h_unp, (size_t)pos_b, hipMemcpyHostToDevice, st_));
  if (fixmode) d_ndot_.upload(h_ndot, st_);
  d_okbits0_.alloc(sizeof(uint32_t) * bits_b);
  d_okbits1_.alloc(sizeof(uint32_t) * bits_b);
  d_ncanon_.alloc(sizeof(int32_t) * n);
  DevBuf& d_plans_all = d_plans_all_;
  d_plans_all.upload(h_plans_, st_);
  BatchArrays b;
  b.seq = d_seq_.as<uint8_t>(); b.ws = d_ws_.as<double>(); b.unp = d_unp_.as<uint8_t>(); b.ndot = nullptr;
  const int min_span = (flags_ & ELEMDP_DBG_NO_TURN) ? 1 : 5;
  // ---- canonical mask (also counts the possible pairs = denominator of bpp_eff)
  HIP_OK(launch_mask(b, d_plans_all.as<SeqPlan>(), n, min_span, !fixmode && !no_rss, d_okbits0_.as<uint32_t>(),
                     d_ncanon_.as<int32_t>(), st_));
  std::vector<int32_t> ncanon(n);
  HIP_OK(hipMemcpyAsync(ncanon.data(), d_ncanon_.as<void>(), sizeof(int32_t) * n, hipMemcpyDeviceToHost, st_));
  HIP_OK(hipStreamSynchronize(st_));
  for (int k = 0; k < n; ++k) h_plans_[k].n_canonical = ncanon[k];
  dbg_lap("load: uploads + canonical mask");
  h_lnbpp_.clear();
  h_lnbpp_base_.clear();

  const uint32_t* final_bits = d_okbits0_.as<uint32_t>();
  const uint32_t* preset_bits = preset_bits_;
  const double* preset_eff = preset_eff_;
  const bool preset = preset_bits && preset_words_ == (size_t)bits_b && !no_rss && !fixmode && min_bpp_ > 0 && !opt_keep_lnbpp_;
  preset_bits_ = nullptr; preset_eff_ = nullptr; preset_words_ = 0;
  bits_words_ = (size_t)bits_b;
  if (preset) {          // the filter's result from an earlier load of the same records (streamed batch)
    final_bits = d_okbits1_.as<uint32_t>();
    HIP_OK(hipMemcpyAsync(d_okbits1_.as<void>(), preset_bits, sizeof(uint32_t) * bits_b, hipMemcpyHostToDevice, st_));
    HIP_OK(hipStreamSynchronize(st_));
    for (int k = 0; k < n; ++k) h_plans_[k].bpp_eff = preset_eff[k];
    dbg_lap("load: filtered mask from the cache");
  } else if (no_rss) {
    HIP_OK(hipMemsetAsync(d_okbits0_.as<void>(), 0, sizeof(uint32_t) * bits_b, st_));
    for (auto& p : h_plans_) p.bpp_eff = 0.;  // em.set_seq is never called in --no-rss mode (motif_model.hpp:57)
  } else if (fixmode) {
    HIP_OK(hipMemcpyAsync(d_okbits0_.as<void>(), h_bits.data(), sizeof(uint32_t) * bits_b, hipMemcpyHostToDevice, st_));
    HIP_OK(hipStreamSynchronize(st_));
    for (int k = 0; k < n; ++k) {
      int nbp = 0;
      const SeqPlan& p = h_plans_[k];
      const int nword = (int)(((int64_t)(p.L + 1) * (p.W + 1) + 31) / 32);
      for (int w = 0; w < nword; ++w) nbp += __builtin_popcount(h_bits[p.bits_base + w]);
      h_plans_[k].bpp_eff = (double)nbp / (double)ncanon[k];
    }
  } else if (min_bpp_ > 0) {
    final_bits = d_okbits1_.as<uint32_t>();
    if (opt_keep_lnbpp_) h_lnbpp_base_.assign(n + 1, 0);
    if (Wmax_ <= kBppLinMaxSpan && !opt_bpp_log_) {
      // ---- K1: BPP filter in the linear semiring (bpp_kernels.hip): no plan of the unfiltered mask; chunks by table memory
      const int64_t cells_cap = 64LL * 1000 * 1000;     // 23 doubles per cell: ~12 GB per chunk
      int first = 0;
      while (first < n) {
        int count = 0;
        int64_t cells = 0, pos = 0;
        int lmax = 0, wmax = 0;
        std::vector<SeqPlan> hp;
        while (first + count < n) {
          SeqPlan p = h_plans_[first + count];
          const int64_t nc = (int64_t)(p.L + 1) * (p.W + 1);
          if (count > 0 && cells + nc > cells_cap) break;
          p.cell_base = cells; p.dmin_base = pos;
          cells += nc; pos += p.L + 1;
          lmax = std::max(lmax, (int)p.L); wmax = std::max(wmax, (int)p.W);
          hp.push_back(p);
          ++count;
        }
        d_bpp_plans_.upload(hp, st_);
        d_bpp_xw_.alloc(sizeof(double) * 5 * (size_t)cells, true);
        // (the per-sequence sweeps read plane entries outside a sequence's triangle with coefficient 0: they must be finite, so a
        // fresh allocation is cleared once; later loads leave finite values of theirs)
        for (auto pb : {std::make_pair(&d_bpp_band_in_, sizeof(double) * kBppInPlanes * (size_t)cells),
                        std::make_pair(&d_bpp_band_out_, sizeof(double) * kBppOutPlanes * (size_t)cells)}) {
          const void* before = pb.first->as<void>();
          const size_t before_b = pb.first->bytes();
          pb.first->alloc(pb.second, true);
          if (pb.first->as<void>() != before || pb.first->bytes() != before_b || !bpp_planes_finite_)
            HIP_OK(hipMemsetAsync(pb.first->as<void>(), 0, pb.first->bytes(), st_));
        }
        bpp_planes_finite_ = true;
        d_bpp_ext_in_.alloc(sizeof(double) * (size_t)pos, true);
        d_bpp_ext_out_.alloc(sizeof(double) * (size_t)pos, true);
        d_bpp_dmin_.alloc(sizeof(int16_t) * (size_t)pos, true);
        d_bpp_kept_.alloc(sizeof(int32_t) * count, true);
        DevBuf d_lnbpp;
        BppLinArgs a;
        std::memset(&a, 0, sizeof(a));
        a.et = d_et_.as<EnergyTables>();
        a.xet = d_xet_.as<EnergyTables>();
        a.cand = d_bpp_cand_.as<BppCandTable>();
        d_bpp_plist_.alloc(sizeof(int16_t) * (size_t)cells, true);
        d_bpp_poff_.alloc(sizeof(int32_t) * (size_t)count * (wmax + 2), true);
        a.plist = d_bpp_plist_.as<int16_t>(); a.poff = d_bpp_poff_.as<int32_t>(); a.poff_stride = wmax + 2;
        for (int k = 0; k < count; ++k) a.pmax = std::max(a.pmax, ncanon[first + k]);
        a.plans = d_bpp_plans_.as<SeqPlan>();
        a.seq = d_seq_.as<uint8_t>();
        a.okbits = d_okbits0_.as<uint32_t>();
        a.dmin = d_bpp_dmin_.as<int16_t>();
        a.xw = d_bpp_xw_.as<double>(); a.xw_stride = (size_t)cells;
        a.tin = d_bpp_band_in_.as<double>(); a.tout = d_bpp_band_out_.as<double>(); a.t_stride = (size_t)cells;
        a.lo_in = d_bpp_ext_in_.as<double>(); a.lo_out = d_bpp_ext_out_.as<double>();
        a.no_ene = (flags_ & ELEMDP_NO_ENERGY) ? 1 : 0;
        a.min_span = min_span;
        a.m_min = (flags_ & ELEMDP_DBG_NO_TURN) ? 4 : 10;
        a.okbits_out = d_okbits1_.as<uint32_t>();
        a.kept = d_bpp_kept_.as<int32_t>();
        a.log_min_bpp = std::log(min_bpp_);
        DevBuf d_prof;
        if (getenv("ELEMDP_BPP_PROF")) {
          d_prof.alloc(sizeof(unsigned long long) * 16);
          HIP_OK(hipMemsetAsync(d_prof.as<void>(), 0, sizeof(unsigned long long) * 16, st_));
          a.prof = d_prof.as<unsigned long long>();
        }
        if (opt_keep_lnbpp_) { d_lnbpp.alloc(sizeof(double) * cells); a.lnbpp = d_lnbpp.as<double>(); }
        HIP_OK(launch_bpp_lin(a, count, lmax, wmax, st_));
        std::vector<int32_t> kept(count);
        HIP_OK(hipMemcpyAsync(kept.data(), d_bpp_kept_.as<void>(), sizeof(int32_t) * count, hipMemcpyDeviceToHost, st_));
        HIP_OK(hipStreamSynchronize(st_));
        for (int k = 0; k < count; ++k) h_plans_[first + k].bpp_eff = (double)kept[k] / (double)ncanon[first + k];
        dbg_lap("load: BPP filter, linear (chunk)");
        if (a.prof) {
          unsigned long long h[16];
          HIP_OK(hipMemcpy(h, a.prof, sizeof(h), hipMemcpyDeviceToHost));
          static const char* nm[8] = {"stage", "stems", "loops generic", "loops 1xn / bulge", "special shapes", "barrier 1", "unary", "barrier 2"};
          for (int dir = 0; dir < 2; ++dir) {
            unsigned long long tot = 0;
            for (int k = 0; k < 8; ++k) tot += h[dir * 8 + k];
            for (int k = 0; k < 8; ++k)
              fprintf(stderr, "[elemdp bpp prof] %s %-18s %6.2f %%  %.3g cycles per sequence\n", dir ? "out" : "in ", nm[k],
                      tot ? 100. * (double)h[dir * 8 + k] / (double)tot : 0., (double)h[dir * 8 + k] / count);
          }
        }
        if (opt_keep_lnbpp_) {
          const size_t base = h_lnbpp_.size();
          h_lnbpp_.resize(base + cells);
          HIP_OK(hipMemcpy(h_lnbpp_.data() + base, d_lnbpp.as<void>(), sizeof(double) * cells, hipMemcpyDeviceToHost));
          for (int k = 0; k < count; ++k) h_lnbpp_base_[first + k] = (int64_t)base + hp[k].cell_base;
        }
        first += count;
      }
    } else {
    // ---- K1 in log space over a plan of the unfiltered mask (bands wider than the linear range; option "bpp_log")
    // ---- K1: BPP filter, in chunks (the plan of the unfiltered mask is large and only needed here)
    int64_t cells_cap = 24LL * 1000 * 1000;
    int first = 0;
    PlanSet& tmp = bpp_plan_;   // (one set of buffers for all chunks and loads: DevBuf::alloc keeps what is large enough;
    tmp.inner_only = true;      //  bounded by cells_cap: a few GB)
    while (first < n) {
      int count = 0;
      int64_t cells = 0;
      while (first + count < n && (count == 0 || cells + (int64_t)(h_plans_[first + count].L + 1) * (h_plans_[first + count].W + 1) <= cells_cap)) {
        cells += (int64_t)(h_plans_[first + count].L + 1) * (h_plans_[first + count].W + 1);
        ++count;
      }
      build_planset(tmp, first, count, d_okbits0_.as<uint32_t>());
      dbg_lap("load: plan of the unfiltered mask (chunk)");
      // table slots for S = 1, one per sequence of the chunk: buffers of their own, so that the (much larger) slots of the
      // evaluation pipelines survive a load_batch -- the mini-batch training mode loads before every evaluation
      {
        const size_t band1 = (size_t)kNumBandStates * (Wmax_ + 1) * (Lmax_ + 1), ext1 = (size_t)(Lmax_ + 1);
        bpp_planes_finite_ = false;   // (log-space values: log 0 = -inf)
        d_bpp_band_in_.alloc(band1 * count * sizeof(double), true);
        d_bpp_band_out_.alloc(band1 * count * sizeof(double), true);
        d_bpp_ext_in_.alloc(ext1 * count * sizeof(double), true);
        d_bpp_ext_out_.alloc(ext1 * count * sizeof(double), true);
        d_bpp_tmp_.alloc(ext1 * 3 * count * sizeof(double), true);
      }
      std::vector<int32_t> order(count);
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](int a2, int b2) { return tmp.h[a2].L > tmp.h[b2].L; });
      DevBuf& d_order = d_bpp_order_; DevBuf& d_rows = d_bpp_rows_; DevBuf& d_kept = d_bpp_kept_;
      DevBuf d_lnbpp;
      d_order.upload(order, st_);
      const int stride = 10;
      d_rows.alloc(sizeof(double) * stride * count);
      d_kept.alloc(sizeof(int32_t) * count);
      TrArgs a;
      std::memset(&a, 0, sizeof(a));
      a.lay = lay0_;
      a.layp = d_lay0_.as<AutomatonLayout>();
      a.ints = d_ints0_.as<int32_t>();
      a.layp_r = a.layp; a.ints_r = a.ints;
      a.params = d_params0_.as<double>();
      a.no_prf = 1;
      a.m_min = (flags_ & ELEMDP_DBG_NO_TURN) ? 4 : 10;
      a.plans = tmp.d_plans.as<SeqPlan>();
      a.grp = d_order.as<int32_t>();
      a.b.seq = d_seq_.as<uint8_t>(); a.b.ws = d_zero_ws_.as<double>(); a.b.unp = d_unp_.as<uint8_t>(); a.b.ndot = nullptr;
      a.okbits = d_okbits0_.as<uint32_t>();
      a.p = tmp.arrays();
      a.band_in = d_bpp_band_in_.as<double>(); a.band_out = d_bpp_band_out_.as<double>();
      a.ext_in = d_bpp_ext_in_.as<double>(); a.ext_out = d_bpp_ext_out_.as<double>();
      a.band_stride = (size_t)kNumBandStates * (Wmax_ + 1) * (Lmax_ + 1);
      a.ext_stride = (size_t)(Lmax_ + 1);
      a.tmp = d_bpp_tmp_.as<double>();
      a.tmp_stride = a.ext_stride;
      a.seq_out = d_rows.as<double>();
      a.out_stride = stride;
      BppOut o;
      o.okbits_out = d_okbits1_.as<uint32_t>();
      o.kept = d_kept.as<int32_t>();
      o.log_min_bpp = std::log(min_bpp_);
      o.lnbpp = nullptr;
      if (opt_keep_lnbpp_) { d_lnbpp.alloc(sizeof(double) * cells); o.lnbpp = d_lnbpp.as<double>(); }
      HIP_OK(hipMemsetAsync(d_rows.as<void>(), 0, sizeof(double) * stride * count, st_));
      const int Lg = tmp.h[order[0]].L;
      HIP_OK(launch_bpp_group(a, o, count, Lg, std::min(Lg, max_span_), st_));
      std::vector<int32_t> kept(count);
      HIP_OK(hipMemcpyAsync(kept.data(), d_kept.as<void>(), sizeof(int32_t) * count, hipMemcpyDeviceToHost, st_));
      HIP_OK(hipStreamSynchronize(st_));
      for (int k = 0; k < count; ++k) h_plans_[first + k].bpp_eff = (double)kept[k] / (double)ncanon[first + k];
      dbg_lap("load: BPP filter (chunk)");
      if (opt_keep_lnbpp_) {
        const size_t base = h_lnbpp_.size();
        h_lnbpp_.resize(base + cells);
        HIP_OK(hipMemcpy(h_lnbpp_.data() + base, d_lnbpp.as<void>(), sizeof(double) * cells, hipMemcpyDeviceToHost));
        for (int k = 0; k < count; ++k) h_lnbpp_base_[first + k] = (int64_t)base + tmp.h[k].cell_base;
      }
      first += count;
    }
    }
  } else {
    for (auto& p : h_plans_) p.bpp_eff = 1.;  // 0 == min_BPP: nbp = total (energy_model.hpp:249-251)
  }
  // ---- resident plan of the filtered mask
  if (final_bits != d_okbits1_.as<uint32_t>())
    HIP_OK(hipMemcpyAsync(d_okbits1_.as<void>(), d_okbits0_.as<void>(), sizeof(uint32_t) * bits_b, hipMemcpyDeviceToDevice, st_));
  plan_.permuted = true;
  build_planset(plan_, 0, n, d_okbits1_.as<uint32_t>());
  dbg_lap("load: plan of the filtered mask");
  for (int k = 0; k < n; ++k) { plan_.h[k].bpp_eff = h_plans_[k].bpp_eff; plan_.h[k].n_canonical = h_plans_[k].n_canonical; }
  plan_.d_plans.upload(plan_.h, st_);
  for (int k = 0; k < n; ++k) h_plans_[k] = plan_.h[k];
  d_order_.upload(h_order_, st_);
  {
    std::vector<SeqPlan> sorted(n);
    for (int k = 0; k < n; ++k) { sorted[k] = h_plans_[h_order_[k]]; sorted[k].index = h_order_[k]; }
    d_plans_sorted_.upload(sorted, st_);
    HIP_OK(hipStreamSynchronize(st_));
  }
  n_cells_total_ = 0;
  for (auto const& pl : h_plans_) n_cells_total_ += (int64_t)(pl.L + 1) * (pl.W + 1);
  d_xwc_.alloc(sizeof(double) * 10 * (size_t)n_cells_total_);
  d_flagged_.alloc(sizeof(int32_t) * ((size_t)n + 1));
  lin_slots_ = 0;
  out_stride_ = 6 + 2 * au_.n_theta() + 4;
  d_seq_out_.alloc(sizeof(double) * (size_t)out_stride_ * n);
  n_slots_ = 0;
  HIP_OK(hipStreamSynchronize(st_));
  dbg_lap("load: weights / output buffers");
  n_seq_ = n;   // committed: everything above succeeded
  opt_eval_first_ = opt_eval_count_ = 0;   // (an evaluation range belongs to the batch it was set for)
  range_key_[0] = range_key_[1] = -1;
}

// ---- streaming --------------------------------------------------------------------------------------------------------
bool Engine::should_stream(const int32_t* off, int n) {
  if (inner_) return false;
  if (opt_max_resident_ > 0) return n > opt_max_resident_;
  // resident needs per sequence: plan (terms, CSR offsets, items in four orders, their weights) + its share of the table
  // slots; a batch whose plan alone would take more than half of the free device memory is streamed
  double cells = 0;
  for (int k = 0; k < n; ++k) {
    const double L = off[k + 1] - off[k], W = std::min<double>(L, max_span_);
    cells += (L + 1) * (W + 1);
  }
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  const size_t held = d_band_in_.bytes() + d_band_out_.bytes() + d_xwc_.bytes() + d_xwi_.bytes();
  return cells * 600.0 > 0.5 * (double)(free_b + held);
}

void Engine::stream_setup(const uint8_t* seq, const int32_t* off, const uint8_t* qual, const int32_t* qoff, const char* fix, int n) {
  if ((flags_ & ELEMDP_DBG_FIX_RSS) && !fix) throw ArgError("load_batch: ELEMDP_DBG_FIX_RSS needs fix_rss strings");
  h_seq_off_.assign(off, off + n + 1);
  h_qual_off_.assign(qoff, qoff + n + 1);
  h_plans_.assign(n, SeqPlan());
  double cells = 0;
  for (int k = 0; k < n; ++k) {
    const int L = off[k + 1] - off[k];
    if (L <= 0) throw ArgError("load_batch: empty sequence");
    if (qoff[k + 1] - qoff[k] != L + 1) throw ArgError("bad seq format. (quality must have L+1 entries)");
    h_plans_[k].L = L;
    h_plans_[k].W = std::min(L, max_span_);
    h_plans_[k].positive = qual[qoff[k + 1] - 1] == 0;
    cells += (double)(L + 1) * (h_plans_[k].W + 1);
  }
  st_seq_.assign(seq + off[0], seq + off[n]);
  st_qual_.assign(qual + qoff[0], qual + qoff[n]);
  if (fix) st_fix_.assign(fix + off[0], fix + off[n]); else st_fix_.clear();
  st_slot_budget_ = 0;
  if (opt_max_resident_ > 0) st_chunk_ = opt_max_resident_;
  else {   // two inner engines, each with a fifth of the free memory for its plan
    size_t free_b = 0, total_b = 0;
    HIP_OK(hipMemGetInfo(&free_b, &total_b));
    const double per_seq = cells / n * 600.0;
    st_chunk_ = (int)std::max(256.0, std::min((double)n, 0.2 * (double)free_b / per_seq));
    st_slot_budget_ = (size_t)(0.2 * (double)free_b);
  }
  st_rows_.clear();
  {
    const int nchunks = (n + st_chunk_ - 1) / st_chunk_;
    st_mask_.assign(nchunks, {});
    st_eff_.assign(nchunks, {});
    st_have_eff_.assign(nchunks, 0);
  }
  // (the buffers of an earlier resident batch would only stand in the way of the inner engines)
  for (DevBuf* b : {&d_band_in_, &d_band_out_, &d_ext_in_, &d_ext_out_, &d_tmp_, &d_xwc_, &d_xwi_, &d_a_in_, &d_a_out_, &d_tr_ext_})
    b->reset();
  n_slots_ = 0; lin_slots_ = 0;
  streaming_ = true;
  n_seq_ = n;
}

void Engine::stream_subs() {
  for (auto& e : sub_) {
    if (e) { e->slot_budget_ = st_slot_budget_; continue; }
    elemdp_model_desc d = desc_;
    d.pattern = desc_pattern_.c_str();
    d.energy_param = desc_has_par_ ? desc_par_.c_str() : nullptr;
    d.device = device_;
    e.reset(new Engine(d));
    e->inner_ = true;
    e->slot_budget_ = st_slot_budget_;
    for (auto const& kv : opt_log_) e->set_option(kv.first, kv.second);
  }
}

void Engine::stream_load_chunk(int k, Engine& e) {
  const int c0 = k * st_chunk_, c1 = std::min(n_seq_, c0 + st_chunk_);
  std::vector<int32_t> off(c1 - c0 + 1), qoff(c1 - c0 + 1);
  for (int t = c0; t <= c1; ++t) { off[t - c0] = h_seq_off_[t] - h_seq_off_[c0]; qoff[t - c0] = h_qual_off_[t] - h_qual_off_[c0]; }
  const size_t s0 = (size_t)(h_seq_off_[c0] - h_seq_off_[0]), q0 = (size_t)(h_qual_off_[c0] - h_qual_off_[0]);
  const bool cached = k < (int)st_mask_.size() && !st_mask_[k].empty();
  if (cached) e.set_filter_preset(st_mask_[k].data(), st_mask_[k].size(), st_eff_[k].data());
  e.load_batch(st_seq_.data() + s0, off.data(), st_qual_.data() + q0, qoff.data(), st_fix_.empty() ? nullptr : st_fix_.data() + s0, c1 - c0);
  if (!cached && k < (int)st_mask_.size() && st_fix_.empty() && !(flags_ & ELEMDP_NO_RSS) && min_bpp_ > 0) e.filter_result(st_mask_[k], st_eff_[k]);
  // the filter's result of the chunk, for elemdp_batch_bpp_eff (chunks write disjoint ranges)
  for (int t = c0; t < c1; ++t) h_plans_[t].bpp_eff = e.h_plans_[t - c0].bpp_eff;
  if (k < (int)st_have_eff_.size()) st_have_eff_[k] = 1;
}

// work(k, c0, c1, engine): chunk k = sequences [c0, c1) is resident on `engine`; the next chunk loads meanwhile
template <class Work> void Engine::stream_chunks(Work work) {
  stream_subs();
  const int nchunks = (n_seq_ + st_chunk_ - 1) / st_chunk_;
  stream_load_chunk(0, *sub_[0]);
  for (int k = 0; k < nchunks; ++k) {
    Engine& cur = *sub_[k & 1];
    std::exception_ptr err;
    std::thread th;
    if (k + 1 < nchunks)
      th = std::thread([&, k] {
        try { stream_load_chunk(k + 1, *sub_[(k + 1) & 1]); } catch (...) { err = std::current_exception(); }
      });
    try {
      work(k, k * st_chunk_, std::min(n_seq_, (k + 1) * st_chunk_), cur);
    } catch (...) {
      if (th.joinable()) th.join();
      throw;
    }
    if (th.joinable()) th.join();
    if (err) std::rethrow_exception(err);
  }
}

void Engine::stream_train(const double* x, int n_param_in, void* partial, bool device_ptr, bool reduce) {
  if (n_param_in != n_param()) throw ArgError("n_param mismatch");
  // (a range of the batch is evaluated by the resident scaled-linear pipeline only: run_lin_batch; anything else would return
  // sums over the whole batch for it)
  if (opt_eval_count_ > 0 && !(opt_eval_first_ == 0 && opt_eval_count_ == n_seq_))
    throw ArgError("eval_first / eval_count: a streamed batch is evaluated as a whole (load fewer sequences, or raise max_resident)");
  set_theta_from(x);
  const int np = partial_len();
  std::vector<double> total(np, 0.), part(np);
  st_rows_.assign((size_t)5 * n_seq_, 0.);
  last_ms[0] = last_ms[1] = last_ms[2] = 0.;
  stream_chunks([&](int, int c0, int c1, Engine& e) {
    e.train_partial(x, n_param_in, part.data(), false, false);
    for (int t = 0; t < np; ++t) total[t] += part[t];
    e.seq_stats(st_rows_.data() + (size_t)5 * c0, c1 - c0);
    for (int t = c0; t < c1; ++t) h_plans_[t].bpp_eff = e.plans()[t - c0].bpp_eff;
    for (int t = 0; t < 3; ++t) last_ms[t] += e.last_ms[t];
  });
  if (comm_ && reduce) {
    HIP_OK(hipMemcpyAsync(d_partial_.as<void>(), total.data(), sizeof(double) * np, hipMemcpyHostToDevice, st_));
    RCCL_OK(Rccl::get().AllReduce(d_partial_.as<void>(), d_partial_.as<void>(), (size_t)np, Rccl::kDouble, Rccl::kSum, comm_, st_));
    HIP_OK(hipMemcpyAsync(total.data(), d_partial_.as<void>(), sizeof(double) * np, hipMemcpyDeviceToHost, st_));
    HIP_OK(hipStreamSynchronize(st_));
  }
  if (device_ptr) HIP_OK(hipMemcpy(partial, total.data(), sizeof(double) * np, hipMemcpyHostToDevice));
  else std::memcpy(partial, total.data(), sizeof(double) * np);
}

void Engine::stream_scan(const double* x, int n_param_in, elemdp_scan_out* out) {
  if (n_param_in != n_param()) throw ArgError("n_param mismatch");
  if (!out) throw ArgError("scan: null output");
  const int nt = au_.n_theta();
  std::vector<double> en(nt, 0.), en_k(nt);
  last_ms[0] = last_ms[1] = last_ms[2] = 0.;
  stream_chunks([&](int, int c0, int c1, Engine& e) {
    (void)c1;
    const size_t so = (size_t)(h_seq_off_[c0] - h_seq_off_[0]), qo = (size_t)(h_qual_off_[c0] - h_qual_off_[0]);
    elemdp_scan_out o = *out;   // the chunk's slices of the caller's arrays (batch offsets)
    if (o.start) o.start += so;
    if (o.inner) o.inner += so;
    if (o.end) o.end += qo;
    if (o.psihat) o.psihat += so;
    if (o.rss) o.rss += so;
    if (o.ys) o.ys += c0;
    if (o.ye) o.ye += c0;
    if (o.exist_prob) o.exist_prob += c0;
    o.en = out->en ? en_k.data() : nullptr;
    e.scan(x, n_param_in, &o);
    if (out->en) for (int t = 0; t < nt; ++t) en[t] += en_k[t];
    for (int t = 0; t < 3; ++t) last_ms[t] += e.last_ms[t];
  });
  if (out->en) std::copy(en.begin(), en.end(), out->en);
}

// Sequences swept in lockstep: as many as fit in ~55 % of the free device memory (at most 8192), then balanced so that all
// groups of the batch have the same size (a small last group runs at lower efficiency).
int Engine::balanced_group(size_t per_slot_bytes, int n) {
  if (n <= 0) n = n_seq_;
  if (opt_group_ > 0) return opt_group_;
  size_t free_b = 0, total_b = 0;
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  const size_t held = d_band_in_.bytes() + d_band_out_.bytes() + d_ext_in_.bytes() + d_ext_out_.bytes();
  size_t budget = (size_t)((double)(free_b + held) * 0.68);
  if (slot_budget_ > 0) budget = std::min(budget, slot_budget_);
  long cap = (long)(budget / std::max<size_t>(per_slot_bytes, 1));
  cap = std::max(1L, std::min(cap, (long)group_cap_));
  const long n_groups = (n + cap - 1) / cap;
  return (int)((n + n_groups - 1) / n_groups);
}

TrArgs Engine::log_pipeline_args() {
  const int S = au_.S();
  TrArgs a;
  std::memset(&a, 0, sizeof(a));
  a.lay = lay_;
  a.layp = d_lay_.as<AutomatonLayout>();
  a.ints = d_ints_.as<int32_t>();
  a.params = d_params_.as<double>();
  a.no_prf = (flags_ & ELEMDP_NO_PROFILE) ? 1 : 0;
  a.m_min = (flags_ & ELEMDP_DBG_NO_TURN) ? 4 : 10;
  a.no_rss = (flags_ & ELEMDP_NO_RSS) ? 1 : 0;
  a.first_pass_only = opt_first_pass_only_ ? 1 : 0;
  a.lik_ratio = (flags_ & ELEMDP_LIK_RATIO) ? 1 : 0;
  a.schedule = (opt_schedule_ == 1 && linear_ok_ && !opt_first_pass_only_) ? 1 : 0;
  a.layp_r = d_layr_.as<AutomatonLayout>();
  a.ints_r = d_intsr_.as<int32_t>();
  a.plans = plan_.d_plans.as<SeqPlan>();
  a.b.seq = d_seq_.as<uint8_t>(); a.b.ws = d_ws_.as<double>(); a.b.unp = d_unp_.as<uint8_t>(); a.b.ndot = nullptr;
  a.okbits = d_okbits1_.as<uint32_t>();
  a.p = plan_.arrays();
  a.band_in = d_band_in_.as<double>(); a.band_out = d_band_out_.as<double>();
  a.ext_in = d_ext_in_.as<double>(); a.ext_out = d_ext_out_.as<double>();
  a.band_stride = (size_t)kNumBandStates * (Wmax_ + 1) * (Lmax_ + 1) * S;
  a.ext_stride = (size_t)(Lmax_ + 1) * S;
  a.tmp = d_tmp_.as<double>();
  a.tmp_stride = a.ext_stride;
  a.seq_out = d_seq_out_.as<double>();
  a.out_stride = out_stride_;
  return a;
}

void Engine::run_train_batch() {
  if (opt_eval_count_ > 0 && !(opt_eval_first_ == 0 && opt_eval_count_ == n_seq_))
    throw ArgError("eval_first / eval_count: the log-space pipeline (pipeline 3) evaluates the whole batch");
  slot_override_ = opt_group_ > 0 ? opt_group_ : 4096;
  ensure_slots(au_.S(), false, n_seq_);
  slot_override_ = 0;
  TrArgs a = log_pipeline_args();
  tables_linear_ = false;
  HIP_OK(hipMemsetAsync(d_seq_out_.as<void>(), 0, sizeof(double) * (size_t)out_stride_ * n_seq_, st_));
  HIP_OK(hipEventRecord(ev_[1], st_));
  for (int g0 = 0; g0 < n_seq_; g0 += n_slots_) {
    const int G = std::min(n_slots_, n_seq_ - g0);
    a.grp = d_order_.as<int32_t>() + g0;
    const int Lg = h_plans_[h_order_[g0]].L;
    HIP_OK(launch_train_group(a, G, Lg, std::min(Lg, max_span_), st_));
  }
  HIP_OK(hipEventRecord(ev_[2], st_));
  HIP_OK(launch_reduce(d_seq_out_.as<double>(), out_stride_, n_seq_, au_.n_theta(), d_partial_.as<double>(), st_));
}

// The scaled-linear pipeline (lin_kernels.hip); sequences it flags (partition function outside the double range, or a
// structurally empty component) are re-evaluated by the log-space pipeline, which applies the reference's skip rule.
// slots, side buffers and the argument record of the scaled-linear pipeline; returns the balanced group size
void Engine::poison_tables() {
  if (!opt_poison_) return;
  for (DevBuf* b : {&d_band_in_, &d_band_out_, &d_a_in_, &d_a_out_})
    if (b->bytes()) HIP_OK(hipMemsetAsync(b->as<void>(), 0xff, b->bytes(), st_));   // (all-ones bytes: NaN)
}

void Engine::lin_weights(int first, int count) {
  LinWeightArgs w;
  const PlanArrays pa = plan_.arrays();
  w.e_stack = pa.e_stack; w.e_ext = pa.e_ext; w.e_ml = pa.e_ml; w.e_close = pa.e_close; w.e_hp = pa.e_hp;
  w.items = pa.items;
  w.items_inner = pa.items_inner; w.items_left = pa.items_left; w.items_right = pa.items_right;
  w.n_cells = (size_t)n_cells_total_; w.n_items = (size_t)plan_.n_items;
  if (count > 0 && count < n_seq_) {   // (records are planned in batch order: the cells of a range of records are contiguous)
    const SeqPlan& p0 = h_plans_[first];
    const SeqPlan& p1 = h_plans_[first + count - 1];
    w.cell_first = (size_t)p0.cell_base;
    w.cell_count = (size_t)(p1.cell_base + (int64_t)(p1.L + 1) * (p1.W + 1) - p0.cell_base);
  }
  w.params = d_params_.as<double>();
  w.xwc = d_xwc_.as<double>(); w.xwi = nullptr;   // (item weights: computed where the records are staged)
  HIP_OK(launch_lin_weights(w, st_));
}

int Engine::prepare_lin(LinArgs& a, bool sched1, bool dense_too, int n_eval) {
  // schedule 1 sweeps the automaton with the shadow copy of (0,0) (one state more per table row); the scan and schedule 0
  // the plain one.  The slots are sized for the wider row.
  const bool shadow = sched1 && lays_.shadow >= 0;
  const AutomatonLayout& L = shadow ? lays_ : lay_;
  const int S = L.S, Sa = std::max(lay_.S, lays_.S), nap = std::max(lay_.ap_rs, lays_.ap_rs);
  // compact band tables (AutomatonLayout::tab_row doubles per cell); the scan's Viterbi pass sweeps dense tables over the
  // same slots (dense_too)
  const int row = std::max(std::max(lay_.tab_row, lays_.tab_row), dense_too ? kNumBandStates * lay_.S : 0);
  // (an evaluation of a range of the resident batch -- options eval_first / eval_count -- needs slots for that range only)
  const int n_need = n_eval > 0 ? std::min(n_eval, n_seq_) : n_seq_;
  {
    const size_t cells = (size_t)(Wmax_ + 1) * (Lmax_ + 1), ext = (size_t)(Lmax_ + 1);
    slot_override_ = balanced_group((cells * row + ext * Sa) * 2 * sizeof(double) + ext * Sa * 3 * sizeof(double) +
                                    cells * nap * 2 * sizeof(double), n_need);
  }
  ensure_slots(Sa, false, n_need, row);
  slot_override_ = 0;
  // (if the allocation had to shrink, rebalance for the slots we got)
  const int n_groups = (n_need + n_slots_ - 1) / n_slots_;
  const int gsz = (n_need + n_groups - 1) / n_groups;
  if (lin_slots_ != n_slots_) {
    d_zs_.alloc(sizeof(double) * 4 * n_slots_);
    const size_t acell = (size_t)(Wmax_ + 1) * (Lmax_ + 1);
    d_a_in_.alloc(sizeof(double) * acell * nap * n_slots_);
    d_a_out_.alloc(sizeof(double) * acell * nap * n_slots_);
    lin_slots_ = n_slots_;
  }
  std::memset(&a, 0, sizeof(a));
  a.lay = L;
  a.layp = shadow ? d_lays_.as<AutomatonLayout>() : d_lay_.as<AutomatonLayout>();
  a.ints = shadow ? d_intss_.as<int32_t>() : d_ints_.as<int32_t>();
  a.params = d_params_.as<double>();
  a.lin = shadow ? d_lins_.as<double>() : d_lin_.as<double>();
  // (table-driven unary phases: not under FIX_RSS, whose fixed pairs need not be canonical -- the weight tables are indexed
  // by the pair type)
  a.fast = (opt_fast_ && !(flags_ & ELEMDP_DBG_FIX_RSS)) ? 1 : 0;
  a.nblk = opt_nblk_;
  a.no_prf = (flags_ & ELEMDP_NO_PROFILE) ? 1 : 0;
  a.m_min = (flags_ & ELEMDP_DBG_NO_TURN) ? 4 : 10;
  a.no_rss = (flags_ & ELEMDP_NO_RSS) ? 1 : 0;
  a.lik_ratio = (flags_ & ELEMDP_LIK_RATIO) ? 1 : 0;
  a.ext_block = (flags_ & (ELEMDP_DBG_NO_TURN | ELEMDP_DBG_FIX_RSS)) ? 1 : 4;   // (pairs span >= 5 positions unless one of these)
  a.plans = plan_.d_plans.as<SeqPlan>();
  a.b.seq = d_seq_.as<uint8_t>(); a.b.ws = d_ws_.as<double>(); a.b.unp = d_unp_.as<uint8_t>(); a.b.ndot = nullptr;
  a.ews = d_ews_.as<double>();
  a.okbits = d_okbits1_.as<uint32_t>();
  a.p = plan_.arrays();
  a.xwc = d_xwc_.as<double>(); a.xwc_stride = (size_t)n_cells_total_;
  a.xwi = nullptr; a.xwi_stride = 0;
  a.band_in = d_band_in_.as<double>(); a.band_out = d_band_out_.as<double>();
  a.ext_in = d_ext_in_.as<double>(); a.ext_out = d_ext_out_.as<double>();
  a.band_stride = band_stride_;
  a.ext_stride = (size_t)(Lmax_ + 1) * S;
  a.zs = d_zs_.as<double>();
  a.a_in = d_a_in_.as<double>(); a.a_out = d_a_out_.as<double>();
  a.a_stride = (size_t)(Wmax_ + 1) * (Lmax_ + 1) * L.ap_rs;
  a.okbits_end = d_okbits_end_.as<uint32_t>();
  a.lmax = Lmax_;
  a.nword_max = nword_max_;
  a.seq_out = d_seq_out_.as<double>();
  a.out_stride = out_stride_;
  a.schedule = sched1 ? 1 : 0;
  a.flagged = d_flagged_.as<int32_t>();
  a.dbg = opt_dbg_;
  a.prof = nullptr;
  if (opt_profile_) {
    d_prof_.alloc(sizeof(long long) * 16 * 64);
    HIP_OK(hipMemsetAsync(d_prof_.as<void>(), 0, sizeof(long long) * 16 * 64, st_));
    a.prof = d_prof_.as<long long>();
  }
  a.n_stage = (L.n_ints <= 4096 && !(opt_dbg_ & 8)) ? L.n_ints : L.n_small;
  tables_S_ = S;
  return gsz;
}

void Engine::run_lin_batch() {
  const bool sched1 = opt_schedule_ == 1 && linear_ok_ && !opt_first_pass_only_ && lay_.s00 == 0 && lays_.shadow >= 0;
  LinArgs a;
  // the records this evaluation covers (options eval_first / eval_count), in processing order (longest first)
  const bool ranged = opt_eval_count_ > 0 && !(opt_eval_first_ == 0 && opt_eval_count_ == n_seq_);
  const int r0 = ranged ? opt_eval_first_ : 0, n_ev = ranged ? opt_eval_count_ : n_seq_;
  if (r0 < 0 || n_ev <= 0 || r0 + n_ev > n_seq_) throw ArgError("eval_first / eval_count outside the resident batch");
  int gsz = prepare_lin(a, sched1, false, n_ev);
  const int32_t* h_ord = h_order_.data();
  const int32_t* d_ord = d_order_.as<int32_t>();
  const SeqPlan* d_sorted = d_plans_sorted_.as<SeqPlan>();
  if (ranged) {
    if (range_key_[0] != r0 || range_key_[1] != n_ev || (int)h_order_r_.size() != n_ev) {
      h_order_r_.clear();
      for (int idx : h_order_) if (idx >= r0 && idx < r0 + n_ev) h_order_r_.push_back(idx);
      std::vector<SeqPlan> sorted(n_ev);
      for (int k = 0; k < n_ev; ++k) { sorted[k] = h_plans_[h_order_r_[k]]; sorted[k].index = h_order_r_[k]; }
      HIP_OK(hipStreamSynchronize(st_));   // (the previous evaluation may still read the arrays)
      d_order_r_.upload(h_order_r_, st_);
      d_plans_sorted_r_.upload(sorted, st_);
      HIP_OK(hipStreamSynchronize(st_));   // (`sorted` is a local)
      range_key_[0] = r0; range_key_[1] = n_ev;
    }
    h_ord = h_order_r_.data();
    d_ord = d_order_r_.as<int32_t>();
    d_sorted = d_plans_sorted_r_.as<SeqPlan>();
    const int n_groups = (n_ev + n_slots_ - 1) / n_slots_;
    gsz = (n_ev + n_groups - 1) / n_groups;
  }
  HIP_OK(hipMemsetAsync(d_seq_out_.as<double>() + (size_t)r0 * out_stride_, 0, sizeof(double) * (size_t)out_stride_ * n_ev, st_));
  HIP_OK(hipMemsetAsync(d_flagged_.as<void>(), 0, sizeof(int32_t), st_));
  if (opt_det_) {   // deterministic mode: a row of counts per (sequence, block of cells), summed in block order by k4_combine
    a.det = 1;
    a.det_nslot = (Lmax_ + 1 + std::max(1, kThreads / a.lay.S) - 1) / std::max(1, kThreads / a.lay.S) + 1;
    const size_t bytes = sizeof(double) * (size_t)n_seq_ * a.det_nslot * out_stride_;
    d_det_.alloc(bytes);
    HIP_OK(hipMemsetAsync(d_det_.as<void>(), 0, bytes, st_));
    a.det_rows = d_det_.as<double>();
  }
  HIP_OK(hipEventRecord(ev_[1], st_));
  lin_weights(r0, n_ev);
  poison_tables();
  // Two groups at a time, each on its own stream and its own half of the table slots: the serial parts of a
  // group (exterior chains, launch tails) run under the band kernels of the other.  (Not for a handful of sequences,
  // whose tables debug_tables reads, nor under the phase profile.)
  const int ns = (opt_group_streams_ >= 2 && n_ev >= 64 && n_slots_ >= 64 && !opt_profile_) ? std::min(opt_group_streams_, kMaxGroupStreams) : 1;
  const int slots_each = n_slots_ / ns;
  int n_groups = (n_ev + slots_each - 1) / slots_each;
  if (ns > 1) n_groups = ((n_groups + ns - 1) / ns) * ns;      // (every stream the same number of groups)
  const int gsz2 = (ns == 1) ? gsz : (n_ev + n_groups - 1) / n_groups;
  auto shifted = [&](LinArgs x, size_t k) {   // the arguments of a group that uses the slots from k on
    x.band_in += k * x.band_stride; x.band_out += k * x.band_stride;
    x.ext_in += k * x.ext_stride; x.ext_out += k * x.ext_stride;
    x.zs += 4 * k;
    x.a_in += k * x.a_stride; x.a_out += k * x.a_stride;
    return x;
  };
  need_group_streams(ns);
  if (ns > 1) {   // the other streams start behind the weights
    HIP_OK(hipEventRecord(gstart_, st_));
    for (int k = 1; k < ns; ++k) HIP_OK(hipStreamWaitEvent(gs_[k], gstart_, 0));
  }
  int gi = 0;
  for (int g0 = 0; g0 < n_ev; g0 += gsz2, ++gi) {
    const int G = std::min(gsz2, n_ev - g0);
    const int k = gi % ns;
    LinArgs ak = shifted(a, (size_t)k * slots_each);
    ak.grp = d_ord + g0;
    ak.plans_slot = d_sorted + g0;
    const int Lg = h_plans_[h_ord[g0]].L;
    HIP_OK(launch_lin_group(ak, G, Lg, std::min(Lg, max_span_), opt_first_pass_only_, gs_[k]));
  }
  for (int k = 1; k < ns; ++k) {   // ... and the main stream continues behind them
    HIP_OK(hipEventRecord(gdone_[k], gs_[k]));
    HIP_OK(hipStreamWaitEvent(st_, gdone_[k], 0));
  }
  int32_t n_flagged = 0;
  HIP_OK(hipMemcpyAsync(&n_flagged, d_flagged_.as<void>(), sizeof(int32_t), hipMemcpyDeviceToHost, st_));
  HIP_OK(hipStreamSynchronize(st_));
  n_flagged_last_ = n_flagged;
  if (opt_profile_) {
    std::vector<long long> hp(16 * 64);
    HIP_OK(hipMemcpy(hp.data(), d_prof_.as<void>(), sizeof(long long) * 16 * 64, hipMemcpyDeviceToHost));
    last_prof.assign(16, 0);
    for (size_t k = 0; k < hp.size(); ++k) last_prof[k % 16] += hp[k];
  }
  tables_linear_ = n_flagged == 0;
  if (n_flagged > 0) {
    ensure_sorted_plan();   // (the log-space kernels sum the role lists in list order: kernels.h)
    TrArgs t = log_pipeline_args();
    // (dense tables over the buffers of the compact ones: as many slots as fit, at least one -- ensure_slots)
    const int n_dense = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_slots_, (band_stride_ * (size_t)n_slots_) / t.band_stride));
    for (int g0 = 0; g0 < n_flagged; g0 += n_dense) {
      const int G = std::min(n_dense, n_flagged - g0);
      t.grp = d_flagged_.as<int32_t>() + 1 + g0;
      HIP_OK(launch_train_group(t, G, Lmax_, Wmax_, st_));
    }
  }
  HIP_OK(hipEventRecord(ev_[2], st_));
  HIP_OK(launch_reduce(d_seq_out_.as<double>() + (size_t)r0 * out_stride_, out_stride_, n_ev, au_.n_theta(), d_partial_.as<double>(), st_));
}

void Engine::run_train(bool) {
  // pipeline 4 (default): the scaled-linear batch pipeline (lin_kernels.hip), which hands sequences outside the double range to
  // pipeline 3, the log-space batch pipeline (train_kernels.hip).  (Pipeline 2, the fused per-sequence kernel of round 1, is
  // retired for training; its scan schedule stays as the scan's range fallback.)
  if (opt_pipeline_ == 3 || opt_det_) ensure_sorted_plan();
  if (opt_pipeline_ == 3) { run_train_batch(); return; }
  run_lin_batch();
}

void Engine::train_partial(const double* x, int n_param_in, void* partial, bool device_ptr, bool reduce) {
  require_device();
  DeviceGuard dg(device_);
  if (streaming_) { stream_train(x, n_param_in, partial, device_ptr, reduce); return; }
  if (n_seq_ <= 0 && !(comm_ && reduce)) throw StateError("train_eval before load_batch");
  if (n_param_in != n_param()) throw ArgError("n_param mismatch");
  HIP_OK(hipEventRecord(ev_[0], st_));
  last_x_.assign(x, x + n_param_in);
  upload_params(x, lay_, false);
  if (n_seq_ > 0) {
    run_train(false);
  } else {   // a rank without a share of the batch: zeros into the all-reduce
    HIP_OK(hipMemsetAsync(d_partial_.as<void>(), 0, sizeof(double) * partial_len(), st_));
    HIP_OK(hipEventRecord(ev_[1], st_));
    HIP_OK(hipEventRecord(ev_[2], st_));
  }
  if (comm_ && reduce)   // sum over the ranks, in place, on the engine's stream (RCCL over xGMI)
    RCCL_OK(Rccl::get().AllReduce(d_partial_.as<void>(), d_partial_.as<void>(), (size_t)partial_len(), Rccl::kDouble, Rccl::kSum, comm_, st_));
  HIP_OK(hipMemcpyAsync(partial, d_partial_.as<void>(), sizeof(double) * partial_len(),
                        device_ptr ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st_));
  HIP_OK(hipEventRecord(ev_[3], st_));
  HIP_OK(hipStreamSynchronize(st_));
  float ms_all = 0, ms_k = 0;
  HIP_OK(hipEventElapsedTime(&ms_all, ev_[0], ev_[3]));
  HIP_OK(hipEventElapsedTime(&ms_k, ev_[1], ev_[2]));
  last_ms[0] = ms_all;
  last_ms[1] = ms_k;
  last_ms[2] = (opt_pipeline_ == 4) ? (double)n_flagged_last_ : 0.;
}

void Engine::train_finish(const double* r, double* fn, double* gr, double* sum_eff, int32_t* n_skipped) {
  const int nt = au_.n_theta();
  if (fn) *fn = r[0];
  if (sum_eff) *sum_eff = r[1];
  if (n_skipped) *n_skipped = (int32_t)std::llround(r[3]);
  if (!gr) return;
  const double* ENo = r + 4;
  const double* ENx = r + 4 + nt;
  const double* EHo = r + 4 + 2 * nt;
  const double* EHx = EHo + 2;
  int k = 0;
  if (softmax()) {  // chain rule through the row-wise softmax (motif_trainer.hpp:251-261)
    if ((int)theta_.size() != nt) throw StateError("train_finish before any evaluation");
    for (int row = 0; row < au_.n_rows(); ++row) {
      const int o = au_.row_offset(row), w = au_.row_width(row);
      double tot = 0.;
      for (int c = 0; c < w; ++c) tot += ENo[o + c] - ENx[o + c];
      for (int c = 0; c < w; ++c) {
        const double tmp = ENo[o + c] - ENx[o + c];
        const double p = std::exp(theta_[o + c]);
        gr[k++] = (1 - p) * tmp - p * (tot - tmp);
      }
    }
  } else {
    for (int t = 0; t < nt; ++t) gr[k++] = ENo[t] - ENx[t];
  }
  gr[k++] = EHo[0] - EHx[0];
  gr[k++] = EHo[1] - EHx[1];
}

void Engine::seq_stats(double* out, int n) {
  require_device();
  DeviceGuard dg(device_);
  if (n != n_seq_) throw ArgError("seq_stats: n_seq mismatch");
  if (streaming_) {
    if ((int)st_rows_.size() != 5 * n) throw StateError("seq_stats before train_eval");
    std::copy(st_rows_.begin(), st_rows_.end(), out);
    return;
  }
  std::vector<double> h((size_t)out_stride_ * n);
  HIP_OK(hipMemcpy(h.data(), d_seq_out_.as<void>(), sizeof(double) * h.size(), hipMemcpyDeviceToHost));
  for (int k = 0; k < n; ++k)
    for (int c = 0; c < 5; ++c) out[5 * k + c] = h[(size_t)k * out_stride_ + c];
}

void Engine::debug_tables(double* inside, double* outside, double* inside_o, double* outside_o, double* ENo, double* ENx,
                          double* EH) {
  require_device();
  DeviceGuard dg(device_);
  if (n_seq_ != 1 || streaming_) throw StateError("debug_tables needs a resident batch of exactly one sequence");
  if (n_slots_ < 1) throw StateError("debug_tables before train_eval");
  if (tables_linear_ && opt_fast_ && !last_x_.empty()) {
    // the table-driven train kernels do not store the planes nothing reads (inside B, outside B and 1): the export repeats the
    // evaluation of the one sequence with the generic kernels, which store every plane
    std::vector<double> part(partial_len());
    opt_fast_ = false;
    try { train_partial(last_x_.data(), n_param(), part.data(), false, false); } catch (...) { opt_fast_ = true; throw; }
    opt_fast_ = true;
  }
  const SeqPlan& p = h_plans_[0];
  const int Sref = au_.S(), L = p.L, W = p.W;
  // (a schedule-1 evaluation of the linear pipeline leaves tables with one more state per row: the shadow of (0,0))
  const int S = (tables_linear_ && tables_S_ > 0) ? tables_S_ : Sref;
  const AutomatonLayout& TL = (S == lays_.S && lays_.shadow >= 0 && S != Sref) ? lays_ : lay_;
  const std::vector<int32_t>& TI = (&TL == &lays_) ? intss_ : ints_;
  const size_t cells = (size_t)(W + 1) * (L + 1);
  // the scaled-linear pipeline keeps Boltzmann weights times a power-of-two scale (lin_rules.h) in COMPACT tables
  // (TableView::ld / st, dp_rules.h): columns only for the states that are useful in a plane, nothing for cells that are not
  // parsable in it -- the export reads those as 0 -> log 0
  const bool lin = tables_linear_;
  const size_t band = lin ? cells * TL.tab_row : (size_t)7 * cells * S, ext = (size_t)(L + 1) * S;
  auto fetch = [&](const DevBuf& src, size_t cnt) {
    std::vector<double> h(cnt);
    HIP_OK(hipMemcpy(h.data(), src.as<void>(), sizeof(double) * cnt, hipMemcpyDeviceToHost));
    return h;
  };
  std::vector<double> cum(L + 1, 0.);   // log2 of prod_{p<j} psb[base(p)]
  if (lin) for (int t = 0; t < L; ++t) cum[t + 1] = cum[t] + h_lin_[kLinPl2 + h_seq_[p.seq_base + t]];
  const double ln2 = 0.69314718055994530942, NEGINF = -std::numeric_limits<double>::infinity();
  auto ref_id = [&](int s) { return TI[TL.st_ref + s]; };   // tables are exported in the reference's state order
  auto conv = [&](double v, double scale_log2) { return !lin ? v : (v > 0. ? std::log(v) - scale_log2 * ln2 : NEGINF); };
  // liveness of a cell in a plane (is_parsable, energy_model.hpp:289-338) from the pair mask and dmin of the sequence
  std::vector<uint32_t> bits((cells + 31) / 32 + 1, 0u);
  std::vector<int16_t> dmin(L + 1, 0);
  if (lin) {
    HIP_OK(hipMemcpy(bits.data(), d_okbits1_.as<uint32_t>() + p.bits_base, sizeof(uint32_t) * ((cells + 31) / 32), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(dmin.data(), plan_.arrays().dmin + p.dmin_base, sizeof(int16_t) * (L + 1), hipMemcpyDeviceToHost));
  }
  const int m_min = (flags_ & ELEMDP_DBG_NO_TURN) ? 4 : 10;
  auto pair_ok = [&](int i, int d) {
    if (i < 0 || d < 0 || d > W || i + d > L) return false;
    const size_t c = (size_t)i * (W + 1) + d;
    return ((bits[c >> 5] >> (c & 31)) & 1u) != 0;
  };
  auto left_ok = [&](int i, int d) { return i >= 0 && d >= 0 && d <= W && i + d <= L && dmin[i] > 0 && d >= dmin[i]; };
  auto cell_live = [&](int e, int d, int i) {
    switch (e) {
      case ST_P: return pair_ok(i, d);
      case ST_E: return i > 0 && d + 2 <= W && pair_ok(i - 1, d + 2);
      case ST_M: return 0 < i && i + d < L && d <= W && m_min <= d;
      case ST_B: case ST_1: case ST_2: return left_ok(i, d);
      default: return true;
    }
  };
  // value of entry (e, d, i, s) of a fetched table in either layout
  auto tab = [&](const std::vector<double>& t, int e, int d, int i, int s2) {
    if (!lin) return t[(((size_t)e * (W + 1) + d) * (L + 1) + i) * S + s2];
    const int c = TI[TL.tab_cmap + e * S + s2];
    if (c < 0 || !cell_live(e, d, i)) return 0.;
    if (TL.tab_cell) return t[((size_t)d * (L + 1) + i) * TL.tab_row + TL.tab_cs[e] + c];
    return t[(size_t)TL.tab_cs[e] * cells + ((size_t)d * (L + 1) + i) * TL.tab_rs[e] + c];
  };
  auto reorder = [&](const std::vector<double>& t, const std::vector<double>* plus2, double* dst, bool outside_tab) {  // -> [i][d][e][s]
    for (int i = 0; i <= L; ++i) for (int d = 0; d <= W; ++d) for (int e = 0; e < 7; ++e) for (int s = 0; s < S; ++s) {
      if (s == TL.shadow) continue;
      double sc = (i + d <= L) ? cum[i + d] - cum[i] : 0.;
      if (outside_tab) sc = cum[L] - sc;
      double v = (i + d <= L) ? tab(t, e, d, i, s) : 0.;
      if (plus2 && e == ST_2 && i + d <= L) v += (*plus2)[((size_t)d * (L + 1) + i) * S + s];
      dst[(((size_t)i * (W + 1) + d) * 7 + e) * Sref + ref_id(s)] = (i + d <= L) ? conv(v, sc) : NEGINF;
    }
  };
  if (inside) reorder(fetch(d_band_in_, band), nullptr, inside, false);
  if (outside) {
    std::vector<double> to = fetch(d_band_out_, band), ha;
    if (lin && TL.n_ap > 0 && d_a_out_.bytes() >= sizeof(double) * cells * TL.ap_rs) {
      // the linear pipeline keeps only the direct part (rules 4a, 3a) of the plane-2 outside values; what arrives through
      // rule 2 is HA(k,l,t) = sum_i sum_{p=(s1,t)} outA(i,l,p) 1(i,k,s1) (lin_rules.h) -- added here for the export
      const int nA = TL.n_ap, nAs = TL.ap_rs;
      const std::vector<double> ti = fetch(d_band_in_, band), ao = fetch(d_a_out_, cells * nAs);
      ha.assign(cells * S, 0.);
      for (int d = 0; d <= W; ++d) for (int i = 0; i + d <= L; ++i) for (int q = 0; q < nA; ++q) {
        const int s1 = TI[TL.ap_s1 + q], t = TI[TL.ap_t + q];
        if (!(tab(ti, ST_2, d, i, t) != 0.)) continue;
        double acc = 0.;
        for (int b = 1; d + b <= W && i - b >= 0; ++b)
          if (left_ok(i - b, b))   // 1(i-b, i, .) is parsable; the pair entries of (i-b, d+b) then exist
            acc += ao[((size_t)(d + b) * (L + 1) + (i - b)) * nAs + q] * tab(ti, ST_1, b, i - b, s1);
        ha[((size_t)d * (L + 1) + i) * S + t] += acc;
      }
    }
    reorder(to, ha.empty() ? nullptr : &ha, outside, true);
  }
  if (inside_o) { auto h = fetch(d_ext_in_, ext); for (int j = 0; j <= L; ++j) for (int s = 0; s < S; ++s) if (s != TL.shadow) inside_o[(size_t)j * Sref + ref_id(s)] = conv(h[(size_t)j * S + s], cum[j]); }
  if (outside_o) { auto h = fetch(d_ext_out_, ext); for (int j = 0; j <= L; ++j) for (int s = 0; s < S; ++s) if (s != TL.shadow) outside_o[(size_t)j * Sref + ref_id(s)] = conv(h[(size_t)j * S + s], cum[L] - cum[j]); }
  std::vector<double> o = fetch(d_seq_out_, out_stride_);
  const int nt = au_.n_theta();
  if (ENo) std::copy(o.begin() + 6, o.begin() + 6 + nt, ENo);
  if (ENx) std::copy(o.begin() + 6 + nt, o.begin() + 6 + 2 * nt, ENx);
  if (EH) std::copy(o.begin() + 6 + 2 * nt, o.begin() + 6 + 2 * nt + 4, EH);
}

void Engine::batch_pairs(int idx, uint8_t* kept, double* lnbpp, int cap) {
  require_device();
  DeviceGuard dg(device_);
  if (streaming_) throw StateError("batch_pairs needs a resident batch (the handle streams this one in chunks)");
  if (idx < 0 || idx >= n_seq_) throw ArgError("batch_pairs: bad sequence index");
  const SeqPlan& p = h_plans_[idx];
  const int nc = (p.L + 1) * (p.W + 1);
  if (cap < nc) throw ArgError("batch_pairs: buffer too small");
  const int nword = (nc + 31) / 32;
  std::vector<uint32_t> w(nword);
  HIP_OK(hipMemcpy(w.data(), d_okbits1_.as<uint32_t>() + p.bits_base, sizeof(uint32_t) * nword, hipMemcpyDeviceToHost));
  for (int c = 0; c < nc; ++c) kept[c] = (w[c >> 5] >> (c & 31)) & 1u;
  if (lnbpp) {
    if (h_lnbpp_base_.empty()) throw StateError("ln BPP was not kept (set option keep_lnbpp before load_batch)");
    std::vector<uint32_t> w0(nword);
    HIP_OK(hipMemcpy(w0.data(), d_okbits0_.as<uint32_t>() + p.bits_base, sizeof(uint32_t) * nword, hipMemcpyDeviceToHost));
    for (int c = 0; c < nc; ++c)
      lnbpp[c] = ((w0[c >> 5] >> (c & 31)) & 1u) ? h_lnbpp_[h_lnbpp_base_[idx] + c] : -std::numeric_limits<double>::infinity();
  }
}

void Engine::scan(const double* x, int n_param_in, elemdp_scan_out* out) {
  require_device();
  DeviceGuard dg(device_);
  if (streaming_) { stream_scan(x, n_param_in, out); return; }
  if (n_seq_ <= 0) throw StateError("scan before load_batch");
  if (n_param_in != n_param()) throw ArgError("n_param mismatch");
  // (--no-rss: load_batch cleared the pair mask, so every sweep reduces to the exterior chain = the profile-HMM
  // forward / backward / Viterbi of motif_model.hpp:171-206 under the scanner functors, motif_scanner.hpp:186-214)
  if (!out) throw ArgError("scan: null output");
  upload_params(x, lay_, false);
  const int nt = au_.n_theta(), n = n_seq_, S = au_.S();
  const size_t n_seqpos = (size_t)h_seq_off_[n], n_pos = n_seqpos + n;
  DevBuf d_start, d_end, d_inner, d_psi, d_rss, d_ys, d_ye, d_exist, d_en;
  d_start.alloc(8 * n_seqpos); d_inner.alloc(8 * n_seqpos); d_end.alloc(8 * n_pos);
  d_psi.alloc(4 * n_seqpos); d_rss.alloc(n_seqpos);
  d_ys.alloc(4 * n); d_ye.alloc(4 * n); d_exist.alloc(8 * n); d_en.alloc(8 * (size_t)n * (nt + 1));
  HIP_OK(hipEventRecord(ev_[1], st_));
  // ---- K4 / K5 (the four sum passes, motif_scanner.hpp:186-202) on the scaled-linear batch pipeline
  std::vector<int32_t> flagged;
  bool cyk_done = false;
  const bool sums_on_batch = opt_pipeline_ == 4;
  if (sums_on_batch) {
    LinArgs a;
    dbg_lap("scan: start");
    // a scan is one pass: 1 024 sequences per group run within 4 % of the largest groups and need a third of the table
    // memory (an evaluation loop that already holds larger groups keeps them)
    group_cap_ = std::max(1024, n_slots_);
    const int gsz = prepare_lin(a, false, true);
    group_cap_ = 8192;
    dbg_lap("scan: prepare_lin (table slots)");
    a.scan = 1;
    a.ys = d_ys.as<int32_t>(); a.ye = d_ye.as<int32_t>();
    a.pos_start = d_start.as<double>(); a.pos_inner = d_inner.as<double>(); a.pos_end = d_end.as<double>();
    a.exist = d_exist.as<double>();
    HIP_OK(hipMemsetAsync(d_start.as<void>(), 0, 8 * n_seqpos, st_));
    HIP_OK(hipMemsetAsync(d_inner.as<void>(), 0, 8 * n_seqpos, st_));
    HIP_OK(hipMemsetAsync(d_end.as<void>(), 0, 8 * n_pos, st_));
    HIP_OK(hipMemsetAsync(d_ys.as<void>(), 0, 4 * n, st_));
    HIP_OK(hipMemsetAsync(d_ye.as<void>(), 0, 4 * n, st_));
    HIP_OK(hipMemsetAsync(d_seq_out_.as<void>(), 0, sizeof(double) * (size_t)out_stride_ * n, st_));
    HIP_OK(hipMemsetAsync(d_flagged_.as<void>(), 0, sizeof(int32_t), st_));
    lin_weights();
    poison_tables();
    // trace records of the Viterbi pass: the exterior chain's rows and the traceback stack per table slot (the band targets keep
    // none: scan_rules.h, cyk_retrace)
    const size_t ext = (size_t)(Lmax_ + 1) * S;
    const int stack_stride = 4 * (4 * (Lmax_ + 2));
    d_tr_ext_.alloc(ext * n_slots_ * sizeof(TraceRec));
    d_tr_stack_.alloc((size_t)n_slots_ * stack_stride * sizeof(int32_t));
    dbg_lap("scan: weights + trace slots");
    a.tr_ext = d_tr_ext_.as<TraceRec>();
    a.trace_stack = d_tr_stack_.as<int32_t>(); a.trace_stack_stride = stack_stride;
    a.sc_psihat = d_psi.as<int32_t>(); a.sc_rss = d_rss.as<char>();
    const bool cyk_on_batch = !(opt_dbg_ & 64);
    // two groups at a time (as in run_lin_batch): each on its own stream, with its half of the table slots -- the five exterior
    // chains of a group run under the band kernels of the other
    const int ns = (opt_group_streams_ >= 2 && n >= 128 && n_slots_ >= 128) ? std::min(opt_group_streams_, kMaxGroupStreams) : 1;
    const int slots_each = n_slots_ / ns;
    int n_groups = (n + slots_each - 1) / slots_each;
    if (ns > 1) n_groups = ((n_groups + ns - 1) / ns) * ns;    // (every stream the same number of groups)
    const int gsz2 = (ns == 1) ? gsz : (n + n_groups - 1) / n_groups;
    need_group_streams(ns);
    if (ns > 1) {
      HIP_OK(hipEventRecord(gstart_, st_));
      for (int k = 1; k < ns; ++k) HIP_OK(hipStreamWaitEvent(gs_[k], gstart_, 0));
    }
    int gi = 0;
    for (int g0 = 0; g0 < n; g0 += gsz2, ++gi) {
      const int G = std::min(gsz2, n - g0);
      const int k = gi % ns;
      hipStream_t st = gs_[k];
      LinArgs ak = a;
      ak.band_in += (size_t)k * slots_each * a.band_stride; ak.band_out += (size_t)k * slots_each * a.band_stride;
      ak.ext_in += (size_t)k * slots_each * a.ext_stride; ak.ext_out += (size_t)k * slots_each * a.ext_stride;
      ak.zs += 4 * (size_t)k * slots_each;
      ak.a_in += (size_t)k * slots_each * a.a_stride; ak.a_out += (size_t)k * slots_each * a.a_stride;
      ak.tr_ext += (size_t)k * slots_each * a.ext_stride;
      ak.trace_stack += (size_t)k * slots_each * stack_stride;
      ak.grp = d_order_.as<int32_t>() + g0;
      ak.plans_slot = d_plans_sorted_.as<SeqPlan>() + g0;
      const int Lg = h_plans_[h_order_[g0]].L;
      HIP_OK(launch_lin_scan_group(ak, G, Lg, std::min(Lg, max_span_), 0, st));
      HIP_OK(launch_lin_scan_group(ak, G, Lg, std::min(Lg, max_span_), 1, st));
      if (cyk_on_batch) HIP_OK(launch_cyk_group(ak, G, Lg, std::min(Lg, max_span_), st));   // K6 on the same table slots
    }
    for (int k = 1; k < ns; ++k) {
      HIP_OK(hipEventRecord(gdone_[k], gs_[k]));
      HIP_OK(hipStreamWaitEvent(st_, gdone_[k], 0));
    }
    cyk_done = cyk_on_batch;
    dbg_lap("scan: launches queued");
    int32_t n_flagged = 0;
    HIP_OK(hipMemcpyAsync(&n_flagged, d_flagged_.as<void>(), sizeof(int32_t), hipMemcpyDeviceToHost, st_));
    HIP_OK(hipStreamSynchronize(st_));
    dbg_lap("scan: device done");
    flagged.resize(n_flagged);
    if (n_flagged) HIP_OK(hipMemcpy(flagged.data(), d_flagged_.as<int32_t>() + 1, sizeof(int32_t) * n_flagged, hipMemcpyDeviceToHost));
    n_flagged_last_ = n_flagged;
    tables_linear_ = false;
  }
  // ---- K6 (Viterbi parse + traceback) on the fused kernel; it also runs the whole schedule for the sequences the
  // linear passes flagged (range check) and for pipeline != 4
  int n_blocks;
  if (sums_on_batch) {   // reuse the table and trace slots of the batch pipeline
    n_blocks = std::min(std::min(n_slots_, 2 * n_cu_), n);
  } else {
    ensure_slots(S, true, n);
    lin_slots_ = 0;   // (the table slots were re-allocated with trace tables)
    n_blocks = std::min(n_slots_, n);
  }
  DpArgs d = base_args(lay_, d_ints_.as<int32_t>(), d_params_.as<double>(), plan_, d_okbits1_.as<uint32_t>(), S);
  d.order = d_order_.as<int32_t>();
  d.tr_ext = d_tr_ext_.as<TraceRec>();
  d.trace_stack = d_tr_stack_.as<int32_t>();
  d.trace_stack_stride = 4 * (4 * (Lmax_ + 2));
  d.sc_start = d_start.as<double>(); d.sc_end = d_end.as<double>(); d.sc_inner = d_inner.as<double>();
  d.sc_psihat = d_psi.as<int32_t>(); d.sc_rss = d_rss.as<char>();
  d.sc_ys = d_ys.as<int32_t>(); d.sc_ye = d_ye.as<int32_t>(); d.sc_exist = d_exist.as<double>(); d.sc_en = d_en.as<double>();
  d.lds = lds_layout(lay_, Lmax_, nword_max_, true);
  d.cyk_only = sums_on_batch ? 1 : 0;
  if (!cyk_done) {
    HIP_OK(hipMemsetAsync(d_counter_.as<void>(), 0, sizeof(int32_t), st_));
    HIP_OK(launch_dp(DP_SCAN, d, n_blocks, st_));
  }
  if (!flagged.empty()) {
    d.cyk_only = 0;
    d.order = d_flagged_.as<int32_t>() + 1;
    d.n_seq = (int32_t)flagged.size();
    HIP_OK(hipMemsetAsync(d_counter_.as<void>(), 0, sizeof(int32_t), st_));
    HIP_OK(launch_dp(DP_SCAN, d, std::min(n_blocks, (int)flagged.size()), st_));
  }
  if (!sums_on_batch) n_slots_ = 0;   // scan slots are not reused by the train pipelines
  HIP_OK(hipEventRecord(ev_[2], st_));
  HIP_OK(hipStreamSynchronize(st_));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, ev_[1], ev_[2]));
  last_ms[0] = last_ms[1] = ms;
  last_ms[2] = (double)flagged.size();
  auto get = [&](void* dst, const DevBuf& src, size_t bytes) { if (dst) HIP_OK(hipMemcpy(dst, src.as<void>(), bytes, hipMemcpyDeviceToHost)); };
  get(out->start, d_start, 8 * n_seqpos);
  get(out->inner, d_inner, 8 * n_seqpos);
  get(out->end, d_end, 8 * n_pos);
  get(out->psihat, d_psi, 4 * n_seqpos);
  get(out->rss, d_rss, n_seqpos);
  get(out->ys, d_ys, 4 * n);
  get(out->ye, d_ye, 4 * n);
  get(out->exist_prob, d_exist, 8 * n);
  if (out->en) {  // E[N] summed over the batch in input order (motif_scanner.hpp:253-259)
    for (int t = 0; t < nt; ++t) out->en[t] = 0.;
    std::vector<char> is_flagged(n, 0);
    for (int k : flagged) is_flagged[k] = 1;
    std::vector<double> h((size_t)n * nt), rows;
    if (!sums_on_batch || !flagged.empty()) HIP_OK(hipMemcpy(h.data(), d_en.as<void>(), 8 * h.size(), hipMemcpyDeviceToHost));
    if (sums_on_batch) {
      rows.resize((size_t)out_stride_ * n);
      HIP_OK(hipMemcpy(rows.data(), d_seq_out_.as<void>(), sizeof(double) * rows.size(), hipMemcpyDeviceToHost));
    }
    for (int k = 0; k < n; ++k)
      for (int t = 0; t < nt; ++t)
        out->en[t] += (sums_on_batch && !is_flagged[k]) ? rows[(size_t)k * out_stride_ + 6 + t] : h[(size_t)k * nt + t];
  }
}


// ---- shuffled negatives (host) ------------------------------------------------------------------------------------
// k-let preserving shuffle by a random Euler tour (uShuffle): vertices = distinct (k-1)-lets in order of first
// appearance, edges = consecutive lets; a random arborescence towards the last let (Wilson), the remaining out-edges of
// every vertex permuted, then the walk from the first let.  The calls of rand() -- `rand() % n` in exactly this order --
// are what makes the result identical to the reference's for the same srand() seed.
namespace {
void kmer_shuffle(const uint8_t* s, int l, int k, uint8_t* t) {
  auto rnd = [](int n) { return (int)(static_cast<long>(std::rand()) % n); };
  if (k >= l) { std::copy(s, s + l, t); return; }
  if (k <= 1) {
    std::copy(s, s + l, t);
    for (int i = l - 1; i > 0; --i) std::swap(t[i], t[rnd(i + 1)]);
    return;
  }
  const int n_lets = l - k + 2;
  std::vector<int> let_vertex(n_lets), first_pos;   // vertex id of every let; first position of every vertex
  for (int i = 0; i < n_lets; ++i) {
    int v = -1;
    for (size_t u = 0; u < first_pos.size() && v < 0; ++u)
      if (std::equal(s + first_pos[u], s + first_pos[u] + (k - 1), s + i)) v = (int)u;
    if (v < 0) { v = (int)first_pos.size(); first_pos.push_back(i); }
    let_vertex[i] = v;
  }
  const int nv = (int)first_pos.size(), root = let_vertex[n_lets - 1];
  std::vector<std::vector<int>> succ(nv);
  for (int i = 0; i + 1 < n_lets; ++i) succ[let_vertex[i]].push_back(let_vertex[i + 1]);
  std::vector<char> intree(nv, 0);
  std::vector<int> next(nv, 0);
  intree[root] = 1;
  for (int i = 0; i < nv; ++i) {
    int u = i;
    while (!intree[u]) { next[u] = rnd((int)succ[u].size()); u = succ[u][next[u]]; }
    u = i;
    while (!intree[u]) { intree[u] = 1; u = succ[u][next[u]]; }
  }
  auto permute = [&](std::vector<int>& a, int n) { for (int i = n - 1; i > 0; --i) std::swap(a[i], a[rnd(i + 1)]); };
  for (int i = 0; i < nv; ++i) {
    std::vector<int>& a = succ[i];
    const int n = (int)a.size();
    if (i != root) { std::swap(a[n - 1], a[next[i]]); permute(a, n - 1); }
    else permute(a, n);
  }
  std::copy(s, s + (k - 1), t);
  std::vector<int> walked(nv, 0);
  int u = 0, pos = k - 1;
  while (walked[u] < (int)succ[u].size()) {
    const int v = succ[u][walked[u]];
    t[pos++] = s[first_pos[v] + k - 2];
    ++walked[u];
    u = v;
  }
}
}  // namespace

}  // namespace elemdp

// =================================================================================================
// C ABI
// =================================================================================================
struct elemdp_handle { elemdp::Engine* e; };

namespace {
int fail(const std::exception& ex) {
  elemdp::g_error = ex.what();
  if (dynamic_cast<const elemdp::HipError*>(&ex)) {
    if (elemdp::g_error.find("no HIP device") != std::string::npos) return ELEMDP_ENODEV;
    return ELEMDP_EHIP;
  }
  if (dynamic_cast<const elemdp::StateError*>(&ex)) return ELEMDP_ESTATE;
  if (dynamic_cast<const std::bad_alloc*>(&ex)) return ELEMDP_ENOMEM;
  return ELEMDP_EINVAL;
}
}  // namespace

#define ELEMDP_TRY try {
#define ELEMDP_CATCH                        \
  return ELEMDP_OK;                         \
  }                                         \
  catch (const std::exception& ex) {        \
    return fail(ex);                        \
  }

extern "C" {

const char* elemdp_last_error(void) { return elemdp::g_error.c_str(); }
int elemdp_abi_version(void) { return ELEMDP_ABI_VERSION; }
int elemdp_set_data_dir(const char* dir) {
  elemdp::g_data_dir = dir ? dir : "";
  return ELEMDP_OK;
}

int elemdp_create(const elemdp_model_desc* desc, elemdp_handle** out) {
  ELEMDP_TRY
  if (!desc || !out) throw elemdp::ArgError("elemdp_create: null argument");
  *out = nullptr;
  std::unique_ptr<elemdp::Engine> e(new elemdp::Engine(*desc));
  *out = new elemdp_handle{e.release()};
  ELEMDP_CATCH
}
int elemdp_destroy(elemdp_handle* h) {
  if (h) { delete h->e; delete h; }
  return ELEMDP_OK;
}
int elemdp_n_param(const elemdp_handle* h) { return h ? h->e->n_param() : ELEMDP_EINVAL; }
int elemdp_n_state(const elemdp_handle* h) { return h ? h->e->n_state() : ELEMDP_EINVAL; }
int elemdp_n_node(const elemdp_handle* h) { return h ? h->e->n_node() : ELEMDP_EINVAL; }

int elemdp_initial_params(const elemdp_handle* h, double lambda_init, double* x, int32_t n_param) {
  ELEMDP_TRY
  if (!h || !x || n_param != h->e->n_param()) throw elemdp::ArgError("elemdp_initial_params: bad argument");
  const elemdp::Automaton& au = h->e->automaton();
  int k = 0;
  for (int r = 0; r < au.n_rows(); ++r)
  {
    // log-softmax of an all-zero score row, summed the way ProfileHMM::calc_theta does (profile_hmm.hpp:103-111)
    double tot = -INFINITY;
    for (int c = 0; c < au.row_width(r); ++c) tot = (tot == -INFINITY) ? 0. : tot + std::log1p(std::exp(0. - tot));
    for (int c = 0; c < au.row_width(r); ++c) x[k++] = h->e->softmax() ? 0. : 0. - tot;
  }
  x[k++] = lambda_init;
  x[k++] = lambda_init;
  ELEMDP_CATCH
}
int elemdp_describe(const elemdp_handle* h, char* buf, int32_t cap) {
  if (!h || !buf) return ELEMDP_EINVAL;
  std::string s = h->e->automaton().to_json();
  {   // sizes of the flattened automaton the kernels sweep (pruned as the options say): for inspection and tests
    const elemdp::AutomatonLayout& L = h->e->layout();
    char extra[320];
    std::snprintf(extra, sizeof(extra), ", \"layout\": {\"S\": %d, \"n_ap\": %d, \"n_quad\": %d, \"n_split\": %d, \"n_lane\": %d, \"n_front\": %d, \"fp_ok\": %d, "
                  "\"shadow\": %d, \"n_ints\": %d, \"fast_blob_in\": %d, \"fast_blob_out\": %d}}",
                  L.S, L.n_ap, L.n_quad, L.n_split, L.n_lane, L.n_front, L.fp_ok, L.shadow, L.n_ints, L.fb_in_n, L.fb_out_n);
    const size_t close = s.rfind('}');
    if (close != std::string::npos) s = s.substr(0, close) + extra;
  }
  if ((int)s.size() + 1 > cap) return ELEMDP_EINVAL;
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}
int elemdp_set_option(elemdp_handle* h, const char* key, double value) {
  ELEMDP_TRY
  if (!h || !key) throw elemdp::ArgError("elemdp_set_option: null argument");
  h->e->set_option(key, value);
  ELEMDP_CATCH
}

int elemdp_load_batch(elemdp_handle* h, const uint8_t* seq_codes, const int32_t* seq_off, const uint8_t* qual,
                      const int32_t* qual_off, const char* fix_rss, int32_t n_seq) {
  ELEMDP_TRY
  if (!h) throw elemdp::ArgError("null handle");
  h->e->load_batch(seq_codes, seq_off, qual, qual_off, fix_rss, n_seq);
  ELEMDP_CATCH
}
int elemdp_batch_bpp_eff(elemdp_handle* h, double* bpp_eff, int32_t n_seq) {
  ELEMDP_TRY
  if (!h || !bpp_eff || n_seq != h->e->n_seq()) throw elemdp::ArgError("elemdp_batch_bpp_eff: bad argument");
  if (!h->e->bpp_eff_known()) throw elemdp::StateError("elemdp_batch_bpp_eff: the batch is streamed in chunks and not every chunk has been loaded yet (evaluate or scan first)");
  for (int k = 0; k < n_seq; ++k) bpp_eff[k] = h->e->plans()[k].bpp_eff;
  ELEMDP_CATCH
}
int elemdp_batch_pairs(elemdp_handle* h, int32_t seq_index, uint8_t* kept, double* lnbpp, int32_t cap) {
  ELEMDP_TRY
  if (!h || !kept) throw elemdp::ArgError("elemdp_batch_pairs: null argument");
  h->e->batch_pairs(seq_index, kept, lnbpp, cap);
  ELEMDP_CATCH
}

int elemdp_partial_len(const elemdp_handle* h) { return h ? h->e->partial_len() : ELEMDP_EINVAL; }
int elemdp_train_partial(elemdp_handle* h, const double* x, int32_t n_param, void* partial, int32_t partial_is_device) {
  ELEMDP_TRY
  if (!h || !x || !partial) throw elemdp::ArgError("elemdp_train_partial: null argument");
  h->e->train_partial(x, n_param, partial, partial_is_device != 0);
  ELEMDP_CATCH
}
int elemdp_train_finish(elemdp_handle* h, const double* reduced, double* fn, double* gr, double* sum_eff,
                        int32_t* n_skipped) {
  ELEMDP_TRY
  if (!h || !reduced) throw elemdp::ArgError("elemdp_train_finish: null argument");
  h->e->train_finish(reduced, fn, gr, sum_eff, n_skipped);
  ELEMDP_CATCH
}
int elemdp_set_finish_params(elemdp_handle* h, const double* x, int32_t n_param) {
  ELEMDP_TRY
  if (!h || !x || n_param != h->e->n_param()) throw elemdp::ArgError("elemdp_set_finish_params: bad argument");
  h->e->set_theta_from(x);
  ELEMDP_CATCH
}
int elemdp_train_eval(elemdp_handle* h, const double* x, int32_t n_param, double* fn, double* gr, double* sum_eff,
                      int32_t* n_skipped) {
  ELEMDP_TRY
  if (!h || !x) throw elemdp::ArgError("elemdp_train_eval: null argument");
  std::vector<double> partial(h->e->partial_len());
  h->e->train_partial(x, n_param, partial.data(), false, true);   // (summed over the ranks when a communicator is set)
  h->e->train_finish(partial.data(), fn, gr, sum_eff, n_skipped);
  ELEMDP_CATCH
}
int elemdp_comm_unique_id(void* id_out) {
  ELEMDP_TRY
  if (!id_out) throw elemdp::ArgError("elemdp_comm_unique_id: null argument");
  elemdp::Rccl& r = elemdp::Rccl::get();
  if (!r.ok()) throw elemdp::HipError("no HIP device available for the collective: librccl.so could not be loaded");
  elemdp::Rccl::UniqueId uid;
  RCCL_OK(r.GetUniqueId(&uid));
  std::memcpy(id_out, &uid, sizeof(uid));
  ELEMDP_CATCH
}
int elemdp_comm_init(elemdp_handle* h, int32_t rank, int32_t world, const void* id) {
  ELEMDP_TRY
  if (!h) throw elemdp::ArgError("null handle");
  h->e->comm_init(rank, world, id);
  ELEMDP_CATCH
}
int elemdp_comm_destroy(elemdp_handle* h) {
  ELEMDP_TRY
  if (!h) throw elemdp::ArgError("null handle");
  h->e->comm_destroy();
  ELEMDP_CATCH
}
int elemdp_train_seq_stats(elemdp_handle* h, double* out, int32_t n_seq) {
  ELEMDP_TRY
  if (!h || !out) throw elemdp::ArgError("elemdp_train_seq_stats: null argument");
  h->e->seq_stats(out, n_seq);
  ELEMDP_CATCH
}
int elemdp_debug_tables(elemdp_handle* h, double* inside, double* outside, double* inside_o, double* outside_o,
                        double* ENo, double* ENx, double* EH) {
  ELEMDP_TRY
  if (!h) throw elemdp::ArgError("null handle");
  h->e->debug_tables(inside, outside, inside_o, outside_o, ENo, ENx, EH);
  ELEMDP_CATCH
}
int elemdp_scan(elemdp_handle* h, const double* x, int32_t n_param, elemdp_scan_out* out) {
  ELEMDP_TRY
  if (!h || !x) throw elemdp::ArgError("elemdp_scan: null argument");
  h->e->scan(x, n_param, out);
  ELEMDP_CATCH
}
int elemdp_last_timing(elemdp_handle* h, double* ms, int32_t n) {
  if (!h || !ms) return ELEMDP_EINVAL;
  for (int k = 0; k < n && k < 3; ++k) ms[k] = h->e->last_ms[k];
  return ELEMDP_OK;
}
int elemdp_debug_profile(elemdp_handle* h, double* cycles, int32_t n) {
  if (!h || !cycles) return ELEMDP_EINVAL;
  for (int k = 0; k < n && k < 16; ++k) cycles[k] = k < (int)h->e->last_prof.size() ? (double)h->e->last_prof[k] : 0.;
  return ELEMDP_OK;
}
int elemdp_kmer_shuffle(const uint8_t* codes, int32_t L, int32_t k, int32_t iter_cnt, uint8_t* out) {
  if (!codes || !out || L <= 0) return ELEMDP_EINVAL;
  int cnt = 0;
  for (int i = 0; i < L; ++i) cnt += codes[i] == codes[0];
  std::srand((unsigned)(cnt + iter_cnt));   // motif_trainer.hpp:147
  elemdp::kmer_shuffle(codes, L, k, out);
  return ELEMDP_OK;
}
int elemdp_epoch_permutation(int32_t n, int32_t seed, int32_t* perm) {
  if (n < 0 || (n > 0 && !perm)) return ELEMDP_EINVAL;
  std::vector<int32_t> v(n);
  std::iota(v.begin(), v.end(), 0);
  std::mt19937 m;
  m.seed((unsigned)seed);
  std::shuffle(v.begin(), v.end(), m);   // fastq_io.hpp:117
  std::copy(v.begin(), v.end(), perm);
  return ELEMDP_OK;
}
const char* elemdp_kernel_name(void) { return "k4_out"; }

}  // extern "C"
