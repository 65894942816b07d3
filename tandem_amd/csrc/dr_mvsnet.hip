// dr_mvsnet.hip -- MI355X engine behind the DrMvsnet operator API (C ABI: include/dr_mi355x.h).
//
// Replaces tandem/libdr/dr_mvsnet/src/dr_mvsnet.cpp (libtorch TorchScript interpreter + cuDNN)
// with a fixed launch plan of hand-written gfx950 kernels (conv_mfma.h, mvs_kernels.h):
//   CallAsync   dr_mvsnet.cpp:125-283  -> MvsEngine::stage_inputs + worker thread
//   forward     dr_mvsnet.cpp:285-331  -> MvsEngine::forward (cva_mvsnet.py:98-184 as ~75 launches)
//   GetResult   dr_mvsnet.cpp:95-107   -> drm_get_result
// The threading contract is the reference's: one worker thread, one mutex, two condition variables;
// CallAsync blocks only while the previous input is still unprocessed.
#include <cmath>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is bound with dlopen when a communicator is asked for

#include "conv_mfma.h"
#include "mvs_kernels.h"
#ifdef DR_PARITY_HOOKS  // conv11 + prob in one launch: built twice in round 5, correct, slower than the two-kernel path (profiles/r05_tail.txt): parity build only
#include "tail_kernels.h"
#endif
#include "fn_front.h"
#include "fn_head3.h"

namespace dr {

std::string &last_error_slot() {
  thread_local std::string s;
  return s;
}

// ------------------------------------------------------------------ TDMW blob (tandem_amd/weights.py)
struct HostTensor {
  std::vector<int> dims;
  std::vector<float> data;
};
struct Blob {
  int depth_num[3];
  float ratio[3];
  int view_aggregation, base;
  std::map<std::string, HostTensor> t;
  const HostTensor &at(const std::string &k) const {
    auto it = t.find(k);
    if (it == t.end()) fail(DR_ERR_IO, "weight blob: missing tensor %s", k.c_str());
    return it->second;
  }
};

static Blob load_blob(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) fail(DR_ERR_IO, "cannot open weight blob %s", path);
  Blob b;
  char magic[8];
  auto rd = [&](void *p, size_t n) {
    if (fread(p, 1, n, f) != n) { fclose(f); fail(DR_ERR_IO, "weight blob %s truncated", path); }
  };
  rd(magic, 8);
  if (memcmp(magic, "TDMW0001", 8)) { fclose(f); fail(DR_ERR_IO, "%s is not a TDMW blob", path); }
  rd(b.depth_num, 12); rd(b.ratio, 12); rd(&b.view_aggregation, 4); rd(&b.base, 4);
  uint32_t n;
  rd(&n, 4);
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t ln, nd;
    rd(&ln, 4);
    std::string name(ln, '\0');
    rd(&name[0], ln);
    rd(&nd, 4);
    HostTensor t;
    size_t cnt = 1;
    for (uint32_t k = 0; k < nd; ++k) { uint32_t d; rd(&d, 4); t.dims.push_back((int)d); cnt *= d; }
    t.data.resize(cnt);
    rd(t.data.data(), cnt * 4);
    b.t[name] = std::move(t);
  }
  fclose(f);
  if (b.base != 8) fail(DR_ERR_UNSUPPORTED, "only feature_net_base_channels=8 is supported (got %d)", b.base);
  return b;
}

// ------------------------------------------------------------------ small host math (double)
static void inv4(const double *m, double *o) {  // Gauss-Jordan with partial pivoting
  double a[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = m[4 * i + j]; a[i][4 + j] = i == j; }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (piv != c) for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[piv][j]);
    const double d = a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] /= d;
    for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j]; }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o[4 * i + j] = a[i][4 + j];
}
static void mul4(const double *a, const double *b, double *o) {
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j]; o[4 * i + j] = s; }
}
// world->pixel 4x4 = [K * W2C(3x4); 0 0 0 1]   (module.py:798-804)
static void world_to_pixel(const float *K9, const double *w2c, double *o) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += (double)K9[3 * i + k] * w2c[4 * k + j]; o[4 * i + j] = s; }
  o[12] = w2c[12]; o[13] = w2c[13]; o[14] = w2c[14]; o[15] = w2c[15];
}

// ------------------------------------------------------------------ engine
struct DevTensor {
  float *d = nullptr;
  int D = 0, H = 0, W = 0, C = 0;
  int pad = 0;  // zero border (pixels) around every H x W plane in memory; H and W stay the logical size
  int split = 0;  // 16: stored as C / 16 consecutive (D,H,W,16) sub-tensors (stage 1's cost volume: each 16-channel pass of conv0 then reads whole records)
  size_t n() const { return (size_t)D * (H + 2 * pad) * (W + 2 * pad) * C; }         // floats in memory
  float *interior() const { return d + ((size_t)pad * (W + 2 * pad) + pad) * C; }    // first logical pixel
};

struct Op {
  enum Kind { PREPROCESS, CONV, SKIPUP, PROB, COSTVOL, REGRESS, EDGE, HIST, SCAN, APPLY, BORDERFIX, TAIL, FRONT, HEAD3 } kind;
  const float *p0 = nullptr, *p1 = nullptr, *p3 = nullptr, *p4 = nullptr;
  float *p2 = nullptr;
  int d0 = 0, d1 = 0, d2 = 0;
  std::string name;
  ConvLaunch conv;
#ifdef DR_PARITY_HOOKS
  TailArgs tail{};                        // TAIL only: conv11 + prob in one launch (tail_kernels.h)
#endif
  FrontArgs front{};                      // FRONT only: preprocess + conv0.0 + conv0.1 in one launch (fn_front.h)
  Head3Args head3{};                      // HEAD3 only: FeatureNet's folded stage-3 head in one launch (fn_head3.h)
  std::function<ConvLaunch(int)> replan;  // CONV only: build candidate `rank` of the planner's ranking
  std::string sig;                        // CONV only: layer signature in conv_tuned.h's column order
  int ncand = 0;
  int stage = 0, shift = 0, bits = 0;
  double flops = 0, bytes = 0;
};

// RCCL, bound lazily (drm_comm_*): an engine that never view-shards does not need the library, and inside a PyTorch
// process the same librccl.so.1 that torch.distributed loaded is picked up (one RCCL per process).
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;  // optional (reporting only)
  static Rccl &get() {
    static Rccl r;
    static std::mutex m;
    std::lock_guard<std::mutex> g(m);
    if (!r.lib) {
      // DR_RCCL_LIB: bind this library instead (tests/test_view_shard_gpu.py runs two ranks on one GPU against a
      // shared-memory stand-in, tests/cpp/rccl_stub.cpp: RCCL itself refuses two ranks on one device)
      if (const char *e = getenv("DR_RCCL_LIB")) r.lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
      else
      for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      if (!r.lib) fail(DR_ERR_UNSUPPORTED, "RCCL not found (librccl.so.1): %s", dlerror());
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
      r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
      r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(r.lib, "ncclReduce"));
      r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.lib, "ncclBroadcast"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
      r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
      if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.Reduce || !r.Broadcast || !r.GetErrorString) {
        r.lib = nullptr;
        fail(DR_ERR_UNSUPPORTED, "RCCL: missing symbols in librccl");
      }
    }
    return r;
  }
  void check(ncclResult_t e, const char *what) const {
    if (e != ncclSuccess) fail(DR_ERR_DEVICE, "RCCL %s: %s", what, GetErrorString(e));
  }
};

// The four result maps of a window go to the pinned host block in ONE kernel (16-byte stores over PCIe) instead of four
// copy-engine transfers: 4 x (launch + completion latency) is most of the time those take for 1.2 MB each.
__global__ __launch_bounds__(256) void k_publish4(const float4 *__restrict__ a, const float4 *__restrict__ b, const float4 *__restrict__ c,
                                                  const float4 *__restrict__ d, float4 *__restrict__ host, size_t n4) {
  const size_t nt = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 4 * n4; i += nt) {
    const size_t k = i / n4, j = i - k * n4;
    host[i] = (k == 0 ? a : (k == 1 ? b : (k == 2 ? c : d)))[j];
  }
}

// Every DR_* switch of the engine, read ONCE when the engine is created (nothing on the launch path calls getenv).  The product library reads
// six of them (profiling, printing, tuning knobs: listed in INTEGRATION.md); every switch that selects a superseded kernel generation, the losing side of a
// settled A/B or a forced fallback is read through hook_env(), i.e. only in the parity build (-DDR_PARITY_HOOKS, libdr_mi355x_hooks.so: what the tests
// that compare generations load) -- in the product those members are constants.
struct MvsSwitches {
  static int num(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
  static bool on(const char *name) { return getenv(name) != nullptr; }
  static int hnum(const char *name, int dflt) { const char *e = hook_env(name); return e ? atoi(e) : dflt; }  // parity build only (dr_common.h): the product returns dflt
  static bool hon(const char *name) { return hook_env(name) != nullptr; }
  // ---- read by the product library ("Environment switches"): profiling, printing, tuning knobs
  bool side_stream = !on("DR_MVS_NO_SIDE_STREAM");       // FeatureNet's stage-2/3 heads on a second stream (off: strictly sequential kernels, for profiles)
  int conv_print = num("DR_CONV_PRINT", 0);              // autotune / debug printing
  std::string autotune_only = getenv("DR_AUTOTUNE_ONLY") ? getenv("DR_AUTOTUNE_ONLY") : "";  // tuning: restrict autotune to layers whose name contains this
  int cv_dchunk[3] = {num("DR_CV_DCHUNK1", 0), num("DR_CV_DCHUNK2", 0), num("DR_CV_DCHUNK3", 0)};  // tuning: depth planes per cost-volume workgroup (0: default)
  int prob_zchunk = num("DR_PROB_ZCHUNK", 0);            // tuning: z-march chunk of k_prob2 (0: default)
  int hist_blocks = std::max(1, num("DR_HIST_BLOCKS", 128));  // tuning: workgroups of a histogram level (each flushes its bins with atomics on a few hot addresses)
  // ---- parity build only: the other side of every settled A/B, superseded generations, forced fallbacks (constants in the product)
  int prob_rows = hnum("DR_PROB_ROWS", 0);               // logits per lane of k_prob2 (2 or 4: measured slower; default 1)
  bool costvol_v2 = hon("DR_COSTVOL_V2");                // k_costvol2 (the product's fallback for depth chunks that are not multiples of 4) everywhere
  bool regress_generic = hon("DR_REGRESS_GENERIC");      // k_regress (the product's fallback for other plane counts) everywhere
  bool shard_allreduce = hon("DR_SHARD_ALLREDUCE");      // view shard: round 2's all-reduce form instead of reduce + broadcast
  // CostRegNet's conv11 + prob as ONE launch (tail_kernels.h): both forms are correct (tests/test_tail_gpu.py) and the memory side of the fusion works
  // (the 78.6 MB tensor between the two layers is gone), but the transposed convolution x 1.8 (halo) and the prob stencil share the same issue slots -- fp32
  // MFMAs and vector work serialise on a SIMD -- and nothing overlaps the kernel's memory side: 0.128 / 0.113 ms (matrix pipe) and 0.118 / 0.109 (vector pipe)
  // at stages 2 / 3 against the two-kernel path's 0.100 / 0.096 (profiles/r05_tail.txt)
  int tail_fused = hnum("DR_TAIL_FUSED", 0);             // 0: the two-kernel path; 1: k_tail_m (transposed convolution on the matrix pipe); 2: k_tail (on the vector pipe)
  int tail_qy = hnum("DR_TAIL_QY", 0), tail_zchunk = hnum("DR_TAIL_ZCHUNK", 0);  // k_tail's tile (quad rows: 4, 8, 16, 32) and depth planes per workgroup (0: chosen by size)
  bool fn_front = hnum("DR_FN_FRONT", 1) != 0;           // 1: FeatureNet's first block (u8 -> float, conv0.0, conv0.1) in one launch (k_fn_front); 0: the three launches
  bool fn_head3 = hnum("DR_FN_HEAD3", 1) != 0;           // 1: the folded stage-3 head of FeatureNet (fn.out3a..d) in one launch (k_fn_head3); 0: the four launches
  bool filter_fused = hnum("DR_FILTER_FUSED", 1) != 0;   // 1: the radix select's scans run as the prologue of the kernels that follow them (5 launches); 0: a k_scan launch per level (8)
  bool prob_regress = hnum("DR_PROB_REGRESS", 1) != 0;   // 1: where a stage's planes are one depth chunk of k_prob2 (D = 8), the regression runs in the same launch (k_prob2_regress)
  bool vol_split = !hon("DR_VOL_NO_SPLIT");              // stage 1's 32-channel cost volume as two 16-channel halves (DevTensor::split); off: one (D,h,w,32) tensor
  bool costvol_v1 = hon("DR_COSTVOL_V1");                // round 2's k_costvol on unpadded feature maps
  bool costvol_v3 = hon("DR_COSTVOL_V3");                // k_costvol3 everywhere: also where the product runs k_costvol5 and where DR_CV4_STAGES selects the LDS-staged k_costvol4
  int costvol_cpl = hnum("DR_COSTVOL_CPL", 4) == 8 ? 8 : 4;
  bool prob_v1 = hon("DR_PROB_V1");                      // round 2's k_prob (L1 gathers)
  int prob_block = std::max(64, std::min(256, hnum("DR_PROB_BLOCK", 256) / 64 * 64)), prob_xo = hnum("DR_PROB_XO", 1);
  bool prob_launch_order = hon("DR_PROB_LAUNCH_ORDER"), prob_on_conv = hon("DR_PROB_ON_CONV");
  bool skip_on_conv = hon("DR_SKIP_ON_CONV"), no_skip_fusion = hon("DR_NO_SKIP_FUSION");
  bool out3_folded = hnum("DR_OUT3_FOLDED", 1) != 0;     // 0: FeatureNet's stage-3 head in its literal order (fused-skip kernel)
  bool d2h_copy = hook_env("DR_MVS_D2H") && !strcmp(hook_env("DR_MVS_D2H"), "copy");  // four copy-engine transfers instead of k_publish4
  // k_costvol5's two choices (round 6, profiles/r06_costvol_ab.txt): a sample whose footprint is the previous plane's issues no gathers (0.109 / 0.172 / 0.120 ->
  // 0.084 / 0.150 / 0.117 ms at depth chunks of 4 / 8 / 8 planes; 0.078 / 0.126 / 0.099 in the single-set form); the workgroup tile is four rows of a quarter segment
  // (0.108 -> 0.099 ms at stage 3, 0.126 -> 0.122 at stage 2, nothing at stage 1)
  int cv5_rows = hnum("DR_CV5_ROWS", 0);                 // 0: the product's rule (4 rows); 1 / 4: that tile at every stage
  bool cv5_reuse = hnum("DR_CV5_REUSE", 1) != 0;         // 0: every sample gathers its four taps
  int cv5_abl = hnum("DR_CV5_ABL", 0);                   // measuring hook: k_costvol5 without its gathers (1), stores (2), tap arithmetic (4)
  // k_costvol4 (round 4: source taps staged through LDS -- north_star's "LDS staging of per-pixel feature slices"): bit-identical to
  // k_costvol3 and measured 8-15 % SLOWER (0.121 / 0.163 / 0.105 against 0.106 / 0.150 / 0.099 ms per stage), so it is not in the product
  int cv4_stages = hnum("DR_CV4_STAGES", 0);             // bit s-1 set = stage s builds its cost volume with k_costvol4 where it applies
  int cv4_sp8 = hnum("DR_CV4_SP8", 0);                   // bit s-1 set = 8 planes per k_costvol4 step at stage s (else 4)
};

// One helper thread that takes half of the operator boundary's host copies (the window into the staging block, the result maps out of
// the pinned block): a single core moves them at ~25 GB/s, i.e. 0.27 + 0.2 ms per 640 x 480 x 7 call on the critical path of TANDEM's
// one-window-in-flight loop.  run() hands it a job, wait() returns when the job is done; the caller does its own half in between.
class HostCopier {
 public:
  HostCopier() : th_(&HostCopier::loop, this) {}
  ~HostCopier() {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
    cv_.notify_all();
    th_.join();
  }
  void run(std::function<void()> job) {
    { std::lock_guard<std::mutex> lk(mu_); job_ = std::move(job); busy_ = true; }
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return !busy_; });
    if (!error_.empty()) { std::string e = error_; error_.clear(); fail(DR_ERR_DEVICE, "%s", e.c_str()); }
  }
  void wait_quiet() {
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return !busy_; });
    error_.clear();
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [&] { return busy_ || quit_; });
      if (quit_) return;
      std::function<void()> job = std::move(job_);
      lk.unlock();
      std::string err;
      try { job(); } catch (const std::exception &e) { err = e.what(); }
      lk.lock();
      error_ = err;
      busy_ = false;
      done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::function<void()> job_;
  std::string error_;
  bool busy_ = false, quit_ = false;
  std::thread th_;
};

static const char *conv_kind_name(const ConvLaunch &c) {
  return c.async == 2 ? (c.march.rm ? "rowmarch" : (c.march.wino ? "winomarch" : "march")) : (c.async == 4 ? "wino" : (c.async ? "async" : ""));
}

class MvsEngine {
 public:
  MvsEngine(const char *path, int device) : device_(device), blob_(load_blob(path)) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device) fail(DR_ERR_DEVICE, "DrMvsnet: no HIP device %d (found %d) -- the MI355X path has no CPU fallback", device, n);
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    // Both streams are created HERE, whether the side stream will be used or not: the runtime (ROCm 7.2, four hardware queues by default) gives a new stream the
    // least-shared queue, and with two streams per engine the main streams of four engines land on four different queues (tools/ubench/stream_queues.hip shows
    // the mapping; one stream per engine puts the fourth engine on the third one's queue: 535 instead of 580 depth maps/s, profiles/r06_queues_side_stream.txt).
    DR_HIP(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
    for (auto *e : {&ev_fork_, &ev_feat2_, &ev_feat3_}) DR_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    side_enabled_ = sw_.side_stream;
    std::vector<float> lut(256);
    for (int i = 0; i < 256; ++i) lut[i] = (float)((double)(float)i / 255.0);
    lut_ = consts_.upload(lut);
    DR_HIP(hipHostMalloc((void **)&march_err_, sizeof(int), hipHostMallocDefault));  // pinned, device-visible: k_conv_m raises it when a ring wait gives up
    *march_err_ = 0;
    worker_ = std::thread(&MvsEngine::loop, this);
  }
  ~MvsEngine() {
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_cv_.wait(lk, [&] { return !unprocessed_; });
      running_ = false;
      input_cv_.notify_all();
    }
    worker_.join();
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    if (comm_) { Rccl::get().CommDestroy(comm_); comm_ = nullptr; }
    release();
    for (float *&h : h_out_) if (h) (void)hipHostFree(h);
    if (h_in_) (void)hipHostFree(h_in_);
    if (ev_h2d_) (void)hipEventDestroy(ev_h2d_);
    if (march_err_) (void)hipHostFree(march_err_);
    if (fc_flag_) (void)hipHostFree(fc_flag_);
    if (up_stream_) { (void)hipStreamSynchronize(up_stream_); (void)hipEventDestroy(ev_hits_); (void)hipStreamDestroy(up_stream_); }
    (void)hipStreamSynchronize(side_);
    for (auto e : {ev_fork_, ev_feat2_, ev_feat3_}) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(side_);
    (void)hipStreamDestroy(stream_);
  }

  // --- reference-API surface -------------------------------------------------------------------
  void call_async(int H, int W, int V, int ref, const uint8_t *const *bgrs, const float *K9, const float *const *c2ws,
                  float dmin, float dmax, float disc) {
    check_args(H, W, V, ref, bgrs, K9, c2ws);
    std::unique_lock<std::mutex> lk(mu_);  // held by the worker for the whole forward (dr_mvsnet.cpp:84-92)
    done_cv_.wait(lk, [&] { return !unprocessed_; });
    stage_inputs(H, W, V, ref, bgrs, K9, c2ws, dmin, dmax, disc, true);
    unprocessed_ = true;
    input_cv_.notify_all();
  }
  bool ready() { return !unprocessed_; }
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return !unprocessed_; });
    rethrow_worker_error();
  }
  void get_result(float *depth, float *conf, float *depth_dense, float *conf_dense) {
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return !unprocessed_; });
    rethrow_worker_error();
    if (!has_output_) fail(DR_ERR_PROTOCOL, "Output should be valid. Maybe you called GetResult more than once?");
    const size_t n = (size_t)H_ * W_;
    const float *src = h_out_[out_cur_];
    copier_.run([=] { memcpy(depth_dense, src + 2 * n, n * 4); memcpy(conf_dense, src + 3 * n, n * 4); });  // two maps each
    memcpy(depth, src, n * 4); memcpy(conf, src + n, n * 4);
    copier_.wait();
    has_output_ = false;
  }
  // The same result WITHOUT the 4.9 MB host copy: pointers into the page-locked block the device wrote it to.  Two blocks alternate,
  // so the maps stay valid while the NEXT call is processed and die when the call after that one starts.
  void get_result_view(const float **depth, const float **conf, const float **depth_dense, const float **conf_dense) {
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return !unprocessed_; });
    rethrow_worker_error();
    if (!has_output_) fail(DR_ERR_PROTOCOL, "Output should be valid. Maybe you called GetResult more than once?");
    const size_t n = (size_t)H_ * W_;
    const float *src = h_out_[out_cur_];
    if (depth) *depth = src;
    if (conf) *conf = src + n;
    if (depth_dense) *depth_dense = src + 2 * n;
    if (conf_dense) *conf_dense = src + 3 * n;
    has_output_ = false;
  }

  // --- key-frame feature cache (extension; see build_fn1) ----------------------------------------
  void set_feature_cache(int capacity) {
    if (capacity < 0 || capacity > 64) fail(DR_ERR_ARG, "drm_set_feature_cache: capacity %d out of range (0 = off, up to 64 key frames)", capacity);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return !unprocessed_; });
    if (capacity == fcache_cap_) return;
    fcache_cap_ = capacity;
    if (H_) {  // a configured engine is re-planned: the entries are sized for the window shape
      DR_HIP(hipSetDevice(device_));
      const int H = H_, W = W_, V = V_;
      H_ = W_ = V_ = 0;
      configure(H, W, V);
      has_output_ = false;
    }
  }
  void feature_cache_stats(uint64_t out[6]) {
    std::unique_lock<std::mutex> lk(mu_);
    out[0] = fc_hits_; out[1] = fc_misses_; out[2] = fc_batch_windows_; out[3] = fc_collisions_; out[4] = fn1_ok_ ? 1 : 0; out[5] = (uint64_t)fcache_.size();
  }

  // --- device-resident hooks -------------------------------------------------------------------
  void upload(int H, int W, int V, int ref, const uint8_t *const *bgrs, const float *K9, const float *const *c2ws,
              float dmin, float dmax, float disc) {
    check_args(H, W, V, ref, bgrs, K9, c2ws);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return !unprocessed_; });
    stage_inputs(H, W, V, ref, bgrs, K9, c2ws, dmin, dmax, disc);
    DR_HIP(hipStreamSynchronize(stream_));
  }
  void forward_n(int iters, float *ms) {
    std::unique_lock<std::mutex> lk(mu_);
    require_config();
    DR_HIP(hipSetDevice(device_));
    InFlight windows(device_, false);  // (several engines driven through this hook at once count as windows in flight, like CallAsync's)
    hipEvent_t e0, e1;
    DR_HIP(hipEventCreate(&e0)); DR_HIP(hipEventCreate(&e1));
    DR_HIP(hipEventRecord(e0, stream_));
    for (int i = 0; i < iters; ++i) forward(nullptr);
    DR_HIP(hipEventRecord(e1, stream_));
    DR_HIP(hipStreamSynchronize(stream_));
    float t = 0;
    DR_HIP(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    cache_mismatch_recovered();
    check_march();
    if (ms) *ms = t;
  }
  // Time the first `k` candidates of every convolution layer's plan ranking in place (hipEvents, layer alone on the
  // stream) and keep the fastest.  Different candidates tile and order the fp32 accumulation differently, so results
  // move at the 1e-7 level; a tuned engine is therefore only bit-reproducible against an equally tuned one.
  // Returns the summed layer time before / after (ms).
  void autotune(int k, float *before_ms, float *after_ms) {
    std::unique_lock<std::mutex> lk(mu_);
    require_config();
    if (comm_ && shard_nsrc_) fail(DR_ERR_PROTOCOL, "autotune(): not on a view-sharded engine (tune the unsharded plan)");
    DR_HIP(hipSetDevice(device_));
    hipEvent_t e0, e1;
    DR_HIP(hipEventCreate(&e0)); DR_HIP(hipEventCreate(&e1));
    auto time_launch = [&](const ConvLaunch &c) {
      for (int i = 0; i < 2; ++i) launch_conv(c, stream_);
      DR_HIP(hipEventRecord(e0, stream_));
      for (int i = 0; i < 8; ++i) launch_conv(c, stream_);
      DR_HIP(hipEventRecord(e1, stream_));
      DR_HIP(hipEventSynchronize(e1));
      float t = 0;
      DR_HIP(hipEventElapsedTime(&t, e0, e1));
      return t / 8;
    };
    double t_before = 0, t_after = 0;
    const bool print = sw_.conv_print > 0;
    const char *only = sw_.autotune_only.empty() ? nullptr : sw_.autotune_only.c_str();
    for (Op &o : ops_) {
      if (o.kind != Op::CONV || !o.replan || (only && o.name.find(only) == std::string::npos)) continue;
      const float t0 = time_launch(o.conv);
      float best = t0;
      int best_rank = 0;
      ConvLaunch best_c = o.conv;
      for (int r = 1; r < std::min(k, o.ncand); ++r) {
        const ConvLaunch c = o.replan(r);
        const float t = time_launch(c);
        if (sw_.conv_print > 1)
          fprintf(stderr, "  cand %-12s rank %2d %s<%d,%d,%d> tile %dx%dx%d lds %zu KB grid %u: %.4f ms\n", o.name.c_str(), r, conv_kind_name(c), c.ci, c.ct, c.pt,
                  c.args.TZ, c.args.TY, c.args.TXT * 16, c.lds_bytes >> 10, c.grid.x, t);
        if (t < best * 0.98f) { best = t; best_rank = r; best_c = c; }
      }
      if (print) fprintf(stderr, "autotune %-12s model %.4f ms %s<%d,%d,%d> tile %dx%dx%d -> rank %d %.4f ms %s<%d,%d,%d> tile %dx%dx%d\n", o.name.c_str(), t0,
                         conv_kind_name(o.conv), o.conv.ci, o.conv.ct, o.conv.pt, o.conv.args.TZ, o.conv.args.TY, o.conv.args.TXT * 16, best_rank, best,
                         conv_kind_name(best_c), best_c.ci, best_c.ct, best_c.pt, best_c.args.TZ, best_c.args.TY, best_c.args.TXT * 16);
      if (print && best_rank != 0)
        fprintf(stderr, "TUNED    {%s,   %d, %d, %d, %d, %d, %d, %d},  // %s %.4f -> %.4f ms\n", o.sig.c_str(), best_c.ci, best_c.ct, best_c.pt, best_c.args.TZ,
                best_c.args.TY, best_c.args.TXT, best_c.async == 2 ? (best_c.march.rm ? 3 : (best_c.march.wino ? 5 : 2)) : best_c.async, o.name.c_str(), t0, best);
      o.conv = best_c;
      t_before += t0; t_after += best;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    DR_HIP(hipStreamSynchronize(stream_));
    check_march();
    if (before_ms) *before_ms = (float)t_before;
    if (after_ms) *after_ms = (float)t_after;
    if (fn1_ok_ && !match_fn1()) { fn1_ok_ = false; fc_fast_ = false; }  // (the feature cache's single-view plan follows the batch plan's instances, or stands down)
  }

  // ---- view sharding hooks (SURVEY 8e / BASELINE configs[2]; no reference counterpart) ----
  // Phase p = 0..3 of the uploaded window: [.. cost volume 1] [regularise 1 .. cost volume 2] [.. cost volume 3]
  // [regularise 3 .. edge filter].  Between phases the host sum-reduces "volume<p+1>" over the ranks.
  void set_view_shard(int nsrc_total) {
    std::unique_lock<std::mutex> lk(mu_);
    if (nsrc_total < 0 || nsrc_total > kMaxSrc) fail(DR_ERR_ARG, "set_view_shard: %d source views unsupported (0..%d)", nsrc_total, kMaxSrc);
    shard_nsrc_ = nsrc_total;
  }
  // The view-shard collective inside the engine: with a communicator set, every cost-volume launch of a sharded window
  // is followed -- on the engine's stream, no host round trip -- by an in-place RCCL sum all-reduce of that volume over
  // the ranks, so drm_forward / CallAsync run the whole sharded depth map as one stream of work; FeatureNet's stage-2/3
  // heads on the side stream and the reduce of stage 1 overlap.  (The host-driven protocol of forward_phase stays as
  // the test double: RCCL refuses two ranks on one device, gloo can emulate them.)
  void comm_init(int rank, int world, const void *unique_id) {
    std::unique_lock<std::mutex> lk(mu_);
    if (comm_) fail(DR_ERR_PROTOCOL, "comm_init: a communicator is already set");
    if (!unique_id || rank < 0 || rank >= world) fail(DR_ERR_ARG, "comm_init: bad rank %d of %d", rank, world);
    Rccl &r = Rccl::get();
    DR_HIP(hipSetDevice(device_));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    r.check(r.CommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
    comm_world_ = world;
    comm_rank_ = rank;
  }
  // ranks RCCL itself reports for the engine's communicator (ncclCommCount): what bench.py prints next to the sharded figure, so that a
  // driver record shows how many GPUs the collective really spanned.  0: no communicator; -1: the bound library has no ncclCommCount.
  int comm_count() {
    std::unique_lock<std::mutex> lk(mu_);
    if (!comm_) return 0;
    Rccl &r = Rccl::get();
    if (!r.CommCount) return -1;
    int n = 0;
    r.check(r.CommCount(comm_, &n), "ncclCommCount");
    return n;
  }
  void comm_destroy() {
    std::unique_lock<std::mutex> lk(mu_);
    if (!comm_) return;
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    Rccl::get().CommDestroy(comm_);
    comm_ = nullptr;
    comm_world_ = 0;
  }
  void forward_phase(int phase) {
    std::unique_lock<std::mutex> lk(mu_);
    require_config();
    if (phase < 0 || phase > 3) fail(DR_ERR_ARG, "forward_phase: phase must be 0..3");
    DR_HIP(hipSetDevice(device_));
    std::vector<size_t> cut{0};
    for (size_t i = 0; i < ops_.size(); ++i) if (ops_[i].kind == Op::COSTVOL) cut.push_back(i + 1);
    cut.push_back(ops_.size());
    phase_mode_ = true;  // the caller reduces between phases
    try { forward(nullptr, cut[phase], cut[phase + 1]); } catch (...) { phase_mode_ = false; throw; }
    phase_mode_ = false;
    DR_HIP(hipStreamSynchronize(stream_));
    check_march();
  }
  void device_tensor(const char *name, void **dptr, size_t *n) {
    std::unique_lock<std::mutex> lk(mu_);
    require_config();
    const DevTensor &t = T(name);
    if (dptr) *dptr = t.d;
    if (n) *n = t.n();
  }
  void download(float *depth, float *conf, float *depth_dense, float *conf_dense) {
    std::unique_lock<std::mutex> lk(mu_);
    require_config();
    DR_HIP(hipSetDevice(device_));
    const size_t n = (size_t)H_ * W_ * 4;
    DR_HIP(hipMemcpyAsync(depth, T("depth").d, n, hipMemcpyDeviceToHost, stream_));
    DR_HIP(hipMemcpyAsync(conf, T("confidence").d, n, hipMemcpyDeviceToHost, stream_));
    DR_HIP(hipMemcpyAsync(depth_dense, T("depth3").d, n, hipMemcpyDeviceToHost, stream_));
    DR_HIP(hipMemcpyAsync(conf_dense, T("conf3").d, n, hipMemcpyDeviceToHost, stream_));
    DR_HIP(hipStreamSynchronize(stream_));
    check_march();
  }
  void get_tensor(const char *name, float *out, size_t n_max, size_t *n, int dims[4]) {
    std::unique_lock<std::mutex> lk(mu_);
    require_config();
    DR_HIP(hipSetDevice(device_));
    const DevTensor &t = T(name);
    const size_t logical = (size_t)t.D * t.H * t.W * t.C;
    if (n) *n = logical;
    if (dims) { dims[0] = t.D; dims[1] = t.H; dims[2] = t.W; dims[3] = t.C; }
    if (out) {
      if (logical > n_max) fail(DR_ERR_ARG, "get_tensor(%s): need %zu floats, have %zu", name, logical, n_max);
      DR_HIP(hipStreamSynchronize(stream_));
      check_march();
      if (t.split) {  // split tensor: sub-tensor k, strided into channels [k * split, (k + 1) * split) of the logical (D, H, W, C) block
        const size_t npos = (size_t)t.D * t.H * t.W;
        for (int k = 0; k < t.C / t.split; ++k)
          DR_HIP(hipMemcpy2D(out + (size_t)k * t.split, (size_t)t.C * 4, t.d + (size_t)k * npos * t.split, (size_t)t.split * 4, (size_t)t.split * 4, npos, hipMemcpyDeviceToHost));
      } else if (!t.pad) DR_HIP(hipMemcpy(out, t.d, logical * 4, hipMemcpyDeviceToHost));
      else  // bordered tensor: the caller gets the logical (D, H, W, C) block
        for (int z = 0; z < t.D; ++z)
          DR_HIP(hipMemcpy2D(out + (size_t)z * t.H * t.W * t.C, (size_t)t.W * t.C * 4, t.interior() + (size_t)z * (t.H + 2 * t.pad) * (t.W + 2 * t.pad) * t.C,
                             (size_t)(t.W + 2 * t.pad) * t.C * 4, (size_t)t.W * t.C * 4, t.H, hipMemcpyDeviceToHost));
    }
  }
  void profile(std::string &names, std::vector<float> &ms) {
    std::unique_lock<std::mutex> lk(mu_);
    require_config();
    if (comm_ && shard_nsrc_) fail(DR_ERR_PROTOCOL, "profile(): a view-sharded engine enqueues collectives -- every rank would have to profile in lockstep");
    DR_HIP(hipSetDevice(device_));
    forward(nullptr);  // warm
    std::vector<hipEvent_t> ev(ops_.size() + 1);
    for (auto &e : ev) DR_HIP(hipEventCreate(&e));
    forward(&ev);
    DR_HIP(hipStreamSynchronize(stream_));
    check_march();
    names.clear(); ms.clear();
    for (size_t i = 0; i < ops_.size(); ++i) {
      float t = 0;
      DR_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      ms.push_back(t);
      const Op &o = ops_[i];
      char kn[64] = "misc";
      if (o.kind == Op::CONV && o.conv.async == 2) snprintf(kn, sizeof kn, "k_conv_m<%d,%d,%d,%d,%d,%d,%d>", o.conv.ci, o.conv.nup, o.conv.ct, o.conv.pt, o.conv.fz, o.conv.ncw, o.conv.march.wino);  // rocprofv3's spelling of the instance
      else if (o.kind == Op::CONV && o.conv.async == 4) snprintf(kn, sizeof kn, "k_conv_w<%d,%d,%d>", o.conv.ci, o.conv.ct, o.conv.pt);
      else if (o.kind == Op::CONV && o.conv.async) snprintf(kn, sizeof kn, "k_conv_a<%d,%d,%d>", o.conv.ci, o.conv.ct, o.conv.pt);
      else if (o.kind == Op::CONV && o.conv.bf3) snprintf(kn, sizeof kn, "k_conv_b<%d,%d,%d>", o.conv.ci, o.conv.ct, o.conv.pt);
      else if (o.kind == Op::CONV && o.conv.args.class_loop > 0) snprintf(kn, sizeof kn, "k_conv_c<%d,%d,%d>", o.conv.ci, o.conv.ct, o.conv.pt);
      else if (o.kind == Op::CONV) snprintf(kn, sizeof kn, "k_conv<%d,%d,%d,%d>", o.conv.ci, o.conv.ct, o.conv.pt, o.conv.fz);
      else if (o.kind == Op::COSTVOL) {
        const CostVolArgs &ca = cv_[o.stage - 1];
        const int Cc = 32 >> (o.stage - 1), dch = ca.planes.D >= 8 ? 8 : 4;
        const bool v4 = cv4_applies(o.stage);
        if (v4) snprintf(kn, sizeof kn, "k_costvol4<%d,%d>", Cc, dch);
        else if (cv5_applies(o.stage)) snprintf(kn, sizeof kn, "k_costvol5<%d,%d>", Cc, cv_[o.stage - 1].dchunk);
        else snprintf(kn, sizeof kn, sw_.costvol_v1 ? "k_costvol<%d>" : (sw_.costvol_v2 ? "k_costvol2<%d>" : "k_costvol3<%d>"), Cc);
      }
      else if (o.kind == Op::PROB) {
        if (sw_.prob_v1) snprintf(kn, sizeof kn, "k_prob");
        else if (o.stage >= 1 && o.stage <= 3 && prob_fused_last_[o.stage - 1]) snprintf(kn, sizeof kn, "k_prob2_regress<8>");
        else snprintf(kn, sizeof kn, "k_prob2<%d>", sw_.prob_rows == 2 || sw_.prob_rows == 4 ? sw_.prob_rows : 1);
      }
#ifdef DR_PARITY_HOOKS
      else if (o.kind == Op::TAIL) snprintf(kn, sizeof kn, o.tail.wmf ? "k_tail_m<%d>" : "k_tail<%d>", std::max(3, tail_nout(o.tail.QY, o.tail.QX)));
#endif
      else if (o.kind == Op::REGRESS) snprintf(kn, sizeof kn, "k_regress");
      else if (o.kind == Op::PREPROCESS) snprintf(kn, sizeof kn, "k_preprocess");
      else if (o.kind == Op::FRONT) snprintf(kn, sizeof kn, "k_fn_front");
      else if (o.kind == Op::HEAD3) snprintf(kn, sizeof kn, "k_fn_head3");
      else if (o.kind == Op::SKIPUP) snprintf(kn, sizeof kn, "k_skip_up<%d>", o.stage);
      else if (o.kind == Op::BORDERFIX) snprintf(kn, sizeof kn, "k_out3_border");
      else snprintf(kn, sizeof kn, "k_filter");
      char line[256];
      snprintf(line, sizeof line, "%s\t%s\t%.6e\t%.6e\n", o.name.c_str(), kn, o.flops, o.bytes);
      names += line;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
  }
  void work(double *flops, double *bytes) {
    std::unique_lock<std::mutex> lk(mu_);
    require_config();
    double f = 0, b = 0;
    for (auto &o : ops_) { f += o.flops; b += o.bytes; }
    if (flops) *flops = f;
    if (bytes) *bytes = b;
  }

 private:
  // ---------------------------------------------------------------- plumbing
  void check_args(int H, int W, int V, int ref, const uint8_t *const *bgrs, const float *K9, const float *const *c2ws) {
    if (!bgrs || !K9 || !c2ws) fail(DR_ERR_ARG, "CallAsync: null pointer argument");
    const int vmin = shard_nsrc_ ? 1 : 2;  // a view-shard rank may hold the reference view only
    if (V < vmin || V > kMaxSrc + 1) fail(DR_ERR_ARG, "CallAsync: view_num=%d unsupported (%d..%d)", V, vmin, kMaxSrc + 1);
    if (ref < 0 || ref >= V) fail(DR_ERR_ARG, "CallAsync: ref_index=%d out of range", ref);
    if (H <= 0 || W <= 0 || H % 32 || W % 32) fail(DR_ERR_ARG, "CallAsync: height/width must be positive multiples of 32 (got %dx%d)", H, W);
    for (int i = 0; i < V - 1; ++i) for (int j = i + 1; j < V; ++j)
      if (bgrs[i] == bgrs[j] || c2ws[i] == c2ws[j])
        fail(DR_ERR_ARG, "ERROR: In Call Async passing the same data for index %d and %d", i, j);  // dr_mvsnet.cpp:153-160
  }
  void require_config() { if (!H_) fail(DR_ERR_PROTOCOL, "no window uploaded yet"); }
  void rethrow_worker_error() {
    if (!worker_error_.empty()) { std::string e = worker_error_; worker_error_.clear(); fail(DR_ERR_DEVICE, "%s", e.c_str()); }
  }
  // windows whose kernels are enqueued or running, per device, over all engines of the process (forward() forks onto its side stream only when it is alone)
  static std::atomic<int> &windows_in_flight(int device) {
    static std::atomic<int> n[16];
    return n[device & 15];
  }
  struct InFlight {
    int dev;
    InFlight(int d, bool counted_already) : dev(d) { if (!counted_already) windows_in_flight(dev).fetch_add(1, std::memory_order_relaxed); }
    ~InFlight() { windows_in_flight(dev).fetch_sub(1, std::memory_order_relaxed); }
  };
  void loop() {  // dr_mvsnet.cpp:83-93
    std::unique_lock<std::mutex> lk(mu_);
    while (running_) {
      if (unprocessed_) {
        try {
          DR_HIP(hipSetDevice(device_));
          InFlight window(device_, prelaunched_);  // (counted from here, or from CallAsync's own launch, to the end of this block)
          if (prelaunched_) prelaunched_ = false;  // (CallAsync enqueued this window's forward itself: stage_inputs)
          else forward(nullptr);
          const size_t n = (size_t)H_ * W_ * 4;
          const int blk = out_cur_ ^ 1;  // the block the previous result does NOT live in (drm_get_result_view: that one may still be read)
          float *ho = h_out_[blk];
          if (sw_.d2h_copy) {  // DR_MVS_D2H=copy: the four copy-engine transfers of round 2 (A/B hook)
            DR_HIP(hipMemcpyAsync(ho, T("depth").d, n, hipMemcpyDeviceToHost, stream_));
            DR_HIP(hipMemcpyAsync(ho + n / 4, T("confidence").d, n, hipMemcpyDeviceToHost, stream_));
            DR_HIP(hipMemcpyAsync(ho + 2 * (n / 4), T("depth3").d, n, hipMemcpyDeviceToHost, stream_));
            DR_HIP(hipMemcpyAsync(ho + 3 * (n / 4), T("conf3").d, n, hipMemcpyDeviceToHost, stream_));
          } else {  // (H and W are multiples of 32: whole float4s)
            hipLaunchKernelGGL(k_publish4, dim3(256), dim3(256), 0, stream_, (const float4 *)T("depth").d, (const float4 *)T("confidence").d,
                               (const float4 *)T("depth3").d, (const float4 *)T("conf3").d, (float4 *)h_out_dev_[blk], n / 16);
          }
          DR_HIP(hipStreamSynchronize(stream_));
          if (cache_mismatch_recovered()) {  // a cache hit that was a key collision: the window ran again without the cache; publish THAT result
            hipLaunchKernelGGL(k_publish4, dim3(256), dim3(256), 0, stream_, (const float4 *)T("depth").d, (const float4 *)T("confidence").d,
                               (const float4 *)T("depth3").d, (const float4 *)T("conf3").d, (float4 *)h_out_dev_[blk], n / 16);
            DR_HIP(hipStreamSynchronize(stream_));
          }
          check_march();
          out_cur_ = blk;
          has_output_ = true;
        } catch (const std::exception &e) { worker_error_ = e.what(); }
        unprocessed_ = false;
        done_cv_.notify_all();
      }
      input_cv_.wait(lk, [&] { return unprocessed_ || !running_; });
    }
  }

  DevTensor &T(const std::string &name) {
    auto it = tensors_.find(tprefix_.empty() ? name : tprefix_ + name);
    if (it == tensors_.end()) fail(DR_ERR_ARG, "unknown tensor '%s'", name.c_str());
    return it->second;
  }
  DevTensor &alloc(const std::string &name, int D, int H, int W, int C, int pad = 0) {
    DevTensor t; t.D = D; t.H = H; t.W = W; t.C = C; t.pad = pad;
    t.d = dalloc<float>(t.n());
    if (pad) DR_HIP(hipMemset(t.d, 0, t.n() * 4));  // the border is written once, here; producers only touch the interior
    const std::string key = tprefix_.empty() ? name : tprefix_ + name;  // (the single-view FeatureNet of the feature cache builds into its own names)
    tensors_[key] = t;
    return tensors_[key];
  }
  void release() {
    for (auto &kv : tensors_) (void)hipFree(kv.second.d);
    tensors_.clear();
    ops_.clear();
    ops1_.clear(); fcache_.clear(); fn1_ok_ = false; fc_fast_ = fc_fill_ = false;  // (a new window shape evicts everything: the entries' buffers are in misc_)
    plan_arena_.reset();
    for (void *p : misc_) (void)hipFree(p);
    misc_.clear();
  }

  // ---------------------------------------------------------------- layer construction
  void fold_bn(const std::string &p, int C, std::vector<float> &sc, std::vector<float> &bi) {
    const auto &g = blob_.at(p + ".weight").data, &b = blob_.at(p + ".bias").data;
    const auto &m = blob_.at(p + ".running_mean").data, &v = blob_.at(p + ".running_var").data;
    sc.resize(C); bi.resize(C);
    for (int c = 0; c < C; ++c) {
      const double s = (double)g[c] / std::sqrt((double)v[c] + 1e-5);
      sc[c] = (float)s;
      bi[c] = (float)((double)b[c] - (double)m[c] * s);
    }
  }
  // FeatureNet's skip connection (1x1 conv + bias + nearest-upsampled coarser level) on the streaming kernel k_skip_up;
  // DR_SKIP_ON_CONV=1 keeps it on the MFMA convolution kernel (A/B hook, and the path for other channel counts).
  DevTensor &add_skip(const std::string &opname, const std::string &wname, const DevTensor &in, const std::string &outname, const DevTensor &coarse) {
    const HostTensor &w = blob_.at(wname + ".weight");
    // (measured at 640x480x7: stage 3, Cin = 8: 0.141 -> 0.108 ms; stage 2, Cin = 16, a quarter of the pixels and twice the
    // weights per lane: 0.038 -> 0.073 ms, so that one stays where it was)
    if (!kParityHooks || sw_.skip_on_conv || w.dims[0] != 32 || w.dims[1] != in.C || in.C != 8 || coarse.C != 32 ||
        coarse.H * 2 != in.H || coarse.W * 2 != in.W)
      return add_conv(opname, wname, "", true, false, in, outname, 1, 1, 1, 1, 1, 1, false, CONV_NORMAL, &coarse, 2);
    DevTensor &out = alloc(outname, in.D, in.H, in.W, 32);
    Op o; o.kind = Op::SKIPUP; o.name = opname;
    o.p0 = in.d; o.p1 = plan_arena_->upload(w.data); o.p2 = out.d; o.d0 = in.D; o.d1 = in.H; o.d2 = in.W; o.stage = in.C;
    o.p3 = plan_arena_->upload(blob_.at(wname + ".bias").data); o.p4 = coarse.d;
    o.flops = 2.0 * in.C * 32 * in.n() / in.C; o.bytes = 4.0 * (in.n() + coarse.n() + out.n());
    ops_.push_back(o);
    return out;
  }
  // Adds one convolution layer (possibly several launches) to the plan; returns the output tensor.
  DevTensor &add_conv(const std::string &opname, const std::string &wname, const std::string &bnname, bool conv_bias, bool relu,
                      const DevTensor &in, const std::string &outname, int k3d, int kh, int kw, int sd, int sh, int sw,
                      bool transposed, ConvMode mode, const DevTensor *add, int add_mode, const ConvFuse *fz = nullptr, int out_pad = 0) {
    const HostTensor &w = blob_.at(wname + ".weight");
    ConvLayer L;
    L.transposed = transposed;
    const int c_out = transposed ? w.dims[1] : w.dims[0], c_in_real = transposed ? w.dims[0] : w.dims[1];
    L.Cout = c_out; L.Cin = in.C;
    L.kd = k3d; L.kh = kh; L.kw = kw; L.sd = sd; L.sh = sh; L.sw = sw; L.relu = relu; L.out_pad = out_pad;
    std::vector<float> padded;
    if (c_in_real != in.C) {  // RGB -> RGB0: zero-pad the input-channel axis of the weights
      if (transposed || c_in_real > in.C) fail(DR_ERR_ARG, "%s: channel mismatch", opname.c_str());
      const int taps = k3d * kh * kw;
      padded.assign((size_t)c_out * in.C * taps, 0.f);
      for (int co = 0; co < c_out; ++co) for (int ci = 0; ci < c_in_real; ++ci) for (int t = 0; t < taps; ++t)
        padded[((size_t)co * in.C + ci) * taps + t] = w.data[((size_t)co * c_in_real + ci) * taps + t];
      L.weight = padded.data();
    } else L.weight = w.data.data();
    if (!bnname.empty()) fold_bn(bnname, c_out, L.scale, L.bias);
    else if (conv_bias) L.bias = blob_.at(wname + ".bias").data;
    ConvPlanOut P0;  // dims first
    {
      auto cz = axis_classes(k3d, sd, transposed, in.D), cy = axis_classes(kh, sh, transposed, in.H), cx = axis_classes(kw, sw, transposed, in.W);
      P0.outD = transposed ? in.D * sd : cz[0].npos; P0.outH = transposed ? in.H * sh : cy[0].npos; P0.outW = transposed ? in.W * sw : cx[0].npos;
    }
    DevTensor &out = alloc(outname, P0.outD, P0.outH, P0.outW, c_out, out_pad);
    emit_conv(opname, L, mode, in, out, add ? add->d : nullptr, add_mode, add ? (add_mode == 2 ? add->n() : out.n()) : 0, fz);
    return out;
  }
  // Plans layer L (in -> out, both existing tensors) and appends its launch(es) to the op list.
  void emit_conv(const std::string &opname, const ConvLayer &L, ConvMode mode, const DevTensor &in, DevTensor &out, const float *add_d, int add_mode,
                 size_t add_n, const ConvFuse *fz = nullptr) {
    const int k3d = L.kd, kh = L.kh, kw = L.kw;
    ConvFuse fzc{};
    if (fz) fzc = *fz;
    const bool fused = fz != nullptr;
    ConvPlanOut P = plan_conv(L, mode, in.d, in.D, in.H, in.W, in.C, out.interior(), add_d, add_mode, *plan_arena_, 0, fz, in.split);
    // the autotuner re-plans this layer with another candidate of the cost model's ranking (weights are kept alive)
    auto keep = std::make_shared<std::vector<float>>(L.weight, L.weight + (size_t)L.Cout * L.Cin * k3d * kh * kw);
    ConvLayer Lc = L;
    const float *in_d = in.d;
    float *out_d = out.interior();
    const int iD = in.D, iH = in.H, iW = in.W, iC = in.C, iS = in.split, ncand = P.ncand;
    auto replan = [this, keep, Lc, mode, in_d, iD, iH, iW, iC, iS, out_d, add_d, add_mode, fzc, fused](int rank) mutable {
      Lc.weight = keep->data();
      return plan_conv(Lc, mode, in_d, iD, iH, iW, iC, out_d, add_d, add_mode, *plan_arena_, rank, fused ? &fzc : nullptr, iS).launches.at(0);
    };
    int idx = 0;
    for (auto &cl : P.launches) {
      Op o; o.kind = Op::CONV; o.conv = cl; o.name = opname + (P.launches.size() > 1 ? "." + std::to_string(idx) : "");
      if (P.launches.size() == 1) {
        o.replan = replan; o.ncand = ncand;
        char sig[160];
        snprintf(sig, sizeof sig, "%d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d", L.Cin, L.Cout, k3d, kh, kw, L.sd, L.sh, L.sw, L.transposed ? 1 : (L.up2 ? 1 + L.up2 : 0),
                 (int)mode + (fused ? 8 : 0) + (iS ? 16 : 0), iD, iH, iW);
        o.sig = sig;
      }
      o.flops = cl.flops;
      if (idx == 0) o.bytes = 4.0 * ((fused ? in.n() / in.C * fzc.cin + in.n() / 4 : in.n()) + out.n() + add_n);
      if (fused) o.flops += 2.0 * fzc.cin * in.C * (in.n() / in.C);
      ops_.push_back(o);
      ++idx;
    }
  }
  // FeatureNet's first block: u8 BGR -> RGB0 / 255, conv0.0 (3 -> 8), conv0.1 (8 -> 8), both 3x3 + BN + ReLU (module.py:461-470).  One launch
  // (k_fn_front, fn_front.h) when the weights have that shape; DR_FN_FRONT=0, the bf16x3 mode and any other shape keep the three launches.
  // the fused FeatureNet kernels need 55 KB / 133.5 KB of dynamic LDS per workgroup: plan them only where the device grants that much to one
  // workgroup (on a part with less the launch would be rejected in every forward while the multi-launch forms still exist)
  bool lds_fits(size_t bytes) const {
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device_) != hipSuccess || optin <= 0) {
      (void)hipGetLastError();
      if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device_) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    return (size_t)optin >= bytes;
  }
  DevTensor &front_block(const std::string &fn, int V, int H, int W) {
    const HostTensor &wa = blob_.at(fn + "conv0.0.conv.weight"), &wb = blob_.at(fn + "conv0.1.conv.weight");
    const bool shape_ok = wa.dims.size() == 4 && wa.dims[0] == 8 && wa.dims[1] == 3 && wa.dims[2] == 3 && wa.dims[3] == 3 &&
                          wb.dims.size() == 4 && wb.dims[0] == 8 && wb.dims[1] == 8 && wb.dims[2] == 3 && wb.dims[3] == 3;
    const bool fits32 = (double)V * H * W * 32.0 < 2147483648.0 && ((uintptr_t)d_bgr_ & 3) == 0;  // (the kernel addresses both tensors with 32-bit byte offsets from an aligned base)
    if (!sw_.fn_front || conv_bf3_policy() || !shape_ok || !fits32 || !lds_fits(kFrontLdsBytes)) {
      DevTensor &img = alloc("image", V, H, W, 4);
      { Op o; o.kind = Op::PREPROCESS; o.name = "preprocess"; o.bytes = (double)V * H * W * (3 + 16); ops_.push_back(o); }
      DevTensor &c3a = cbr2("fn.conv0.0", fn + "conv0.0", img, 3, 1, CONV_XPAIR);
      return cbr2("fn.conv0.1", fn + "conv0.1", c3a, 3, 1, CONV_XPAIR);
    }
    DevTensor &c3 = alloc("fn.conv0.1", V, H, W, 8);
    Op o; o.kind = Op::FRONT; o.name = "fn.front";
    FrontArgs &a = o.front;
    a.bgr = d_bgr_; a.lut = lut_; a.out = c3.d;
    a.w1 = reinterpret_cast<const float4 *>(plan_arena_->upload(front_pack(wa.data.data(), 3, 4)));
    a.w2 = reinterpret_cast<const float4 *>(plan_arena_->upload(front_pack(wb.data.data(), 8, 8)));
    auto affine = [&](const std::string &bn) {  // 16 scales, 16 biases: XPAIR row r = 8 * (x of the pair) + channel
      std::vector<float> sc, bi, sb(32);
      fold_bn(bn, 8, sc, bi);
      for (int r = 0; r < 16; ++r) { sb[r] = sc[r & 7]; sb[16 + r] = bi[r & 7]; }
      return plan_arena_->upload(sb);
    };
    a.sb1 = affine(fn + "conv0.0.bn"); a.sb2 = affine(fn + "conv0.1.bn");
    a.V = V; a.H = H; a.W = W;
    a.tilesY = cdiv(H, kFrontTY); a.tilesX = cdiv(W, kFrontTXP); a.ntiles = V * a.tilesY * a.tilesX;
    o.flops = 2.0 * V * H * W * 9.0 * (4 * 8 + 8 * 8);  // (as the planner counts the two layers: RGB0 has four channels)
    o.bytes = (double)V * H * W * (3 + 32);
    ops_.push_back(o);
    return c3;
  }
  DevTensor &cbr2(const std::string &name, const std::string &p, const DevTensor &in, int k, int s, ConvMode m) {
    return add_conv(name, p + ".conv", p + ".bn", false, true, in, name, 1, k, k, 1, s, s, false, m, nullptr, 0);
  }
  DevTensor &cbr3(const std::string &name, const std::string &p, const DevTensor &in, int sd, int shw, ConvMode m) {
    return add_conv(name, p + ".conv", p + ".bn", false, true, in, name, 3, 3, 3, sd, shw, shw, false, m, nullptr, 0);
  }
  DevTensor &dbr3(const std::string &name, const std::string &p, const DevTensor &in, int sd, const DevTensor &skip) {
    return add_conv(name, p + ".conv", p + ".bn", false, true, in, name, 3, 3, 3, sd, 2, 2, true, CONV_NORMAL, &skip, 1);
  }

  // (Re)builds the whole plan for a new window shape.  The engine keeps H_ = W_ = V_ = 0 ("not configured") until the
  // plan is complete: if anything below throws (out of memory, unsupported depth_num, missing tensor) everything
  // allocated so far is released and the next call starts from scratch instead of running a half-built plan.
  void configure(int H, int W, int V) {
    if (H == H_ && W == W_ && V == V_) return;
    DR_HIP(hipStreamSynchronize(stream_));
    release();
    H_ = W_ = V_ = 0;
    memset(cv_, 0, sizeof cv_);
    memset(rg_, 0, sizeof rg_);
    try {
      build_plan(H, W, V);
    } catch (...) {
      release();
      throw;
    }
    H_ = H; W_ = W; V_ = V;
  }
  // FeatureNet for a batch of V views (module.py:461-531): appends its launches to ops_ and records which of them may run on the side stream
  void build_featurenet(int V, int H, int W) {
    const std::string fn = "feature_net.";
    DevTensor &c3 = front_block(fn, V, H, W);
    DevTensor &c2a = cbr2("fn.conv1.0", fn + "conv1.0", c3, 5, 2, CONV_NORMAL);
    DevTensor &c2b = cbr2("fn.conv1.1", fn + "conv1.1", c2a, 3, 1, CONV_NORMAL);
    DevTensor &c2 = cbr2("fn.conv1.2", fn + "conv1.2", c2b, 3, 1, CONV_NORMAL);
    DevTensor &c1a = cbr2("fn.conv2.0", fn + "conv2.0", c2, 5, 2, CONV_NORMAL);
    DevTensor &c1b = cbr2("fn.conv2.1", fn + "conv2.1", c1a, 3, 1, CONV_NORMAL);
    DevTensor &c1 = cbr2("fn.conv2.2", fn + "conv2.2", c1b, 3, 1, CONV_NORMAL);
    // feature maps handed to the cost volume carry a one-pixel zero border (k_costvol2); DR_COSTVOL_V1=1: the unpadded layout + k_costvol
    const int fpad = sw_.costvol_v1 ? 0 : 1;
    add_conv("fn.out1", fn + "out.stage1", "", false, false, c1, "feat1", 1, 1, 1, 1, 1, 1, false, CONV_NORMAL, nullptr, 0, nullptr, fpad);
    fork_lo_ = ops_.size();
    DevTensor &i2 = add_skip("fn.skip2", fn + "skip.stage2", c2, "inter2", c1);
    add_conv("fn.out2", fn + "out.stage2", "", false, false, i2, "feat2", 1, 3, 3, 1, 1, 1, false, CONV_NORMAL, nullptr, 0, nullptr, fpad);
    feat2_op_ = ops_.size() - 1;
    // stage 3: skip.stage3 (1x1, 8 -> 32, + upsampled inter2) is computed inside out.stage3's staging step -- the
    // 32-channel full-resolution tensor between them (275 MB at 640x480x7) is never written or read.  DR_NO_SKIP_FUSION=1: the two-kernel path.
    const HostTensor &w3 = blob_.at(fn + "skip.stage3.weight");
    const HostTensor &wo3 = blob_.at(fn + "out.stage3.weight");
    // out.stage3 is linear in inter3 = up(inter2) + skip.stage3(c3) (no BatchNorm, no ReLU, no bias: module.py:480-485,524-529), so it
    // is evaluated as   conv3x3(c3; Wout . Wskip)  +  conv3x3(up(inter2); Wout)  +  (Wout . bskip, corrected at the image border):
    // an 8 -> 8 XPAIR layer at full resolution (composed weights), two 2 x 3-tap phase layers over inter2 at HALF resolution that
    // add their rows in place (ConvLayer::up2: the upsampled tensor never exists), and a border kernel.  11.0 GFLOP become 6.9, the
    // 275 MB inter3 and the skip staging disappear, and all three convolutions run on the persistent kernels (0.22 -> 0.13 ms).
    // feat3 then differs from the literal order by fp32 reassociation (2e-6 of its range).  DR_OUT3_FOLDED=0: the fused-skip form.
    if (sw_.out3_folded && !sw_.no_skip_fusion && !sw_.skip_on_conv && w3.dims[0] == 32 && w3.dims[1] == 8 && c3.C == 8 &&
        i2.C == 32 && wo3.dims[0] == 8 && wo3.dims[1] == 32 && i2.H * 2 == c3.H && i2.W * 2 == c3.W) {
      const std::vector<float> &b3 = blob_.at(fn + "skip.stage3.bias").data;
      std::vector<float> wa((size_t)8 * 8 * 9), T(9 * 8), bint(8, 0.f);
      for (int co = 0; co < 8; ++co)
        for (int t = 0; t < 9; ++t) {
          for (int c8 = 0; c8 < 8; ++c8) {
            double acc = 0;
            for (int c = 0; c < 32; ++c) acc += (double)wo3.data[((size_t)co * 32 + c) * 9 + t] * (double)w3.data[(size_t)c * 8 + c8];
            wa[((size_t)co * 8 + c8) * 9 + t] = (float)acc;
          }
          double tb = 0;
          for (int c = 0; c < 32; ++c) tb += (double)wo3.data[((size_t)co * 32 + c) * 9 + t] * (double)b3[c];
          T[t * 8 + co] = (float)tb;
        }
      for (int co = 0; co < 8; ++co) { double b = 0; for (int t = 0; t < 9; ++t) b += (double)T[t * 8 + co]; bint[co] = (float)b; }
      DevTensor &f3 = alloc("feat3", c3.D, c3.H, c3.W, 8, fpad);
      if (sw_.fn_head3 && !conv_bf3_policy() && (double)f3.n() * 4.0 < 2147483648.0 && (double)i2.n() * 4.0 < 2147483648.0 && lds_fits(kH3LdsBytes)) {  // one launch: the three terms meet in one accumulator (fn_head3.h)
        Op o; o.kind = Op::HEAD3; o.name = "fn.head3";
        Head3Args &a = o.head3;
        a.c0 = c3.d; a.i2 = i2.d;
        a.wa = reinterpret_cast<const float4 *>(plan_arena_->upload(h3_pack_a(wa.data())));
        a.wb = reinterpret_cast<const float4 *>(plan_arena_->upload(h3_pack_b(wo3.data.data())));
        std::vector<float> b16(16);
        for (int r = 0; r < 16; ++r) b16[r] = bint[r & 7];
        a.bias16 = plan_arena_->upload(b16); a.T = plan_arena_->upload(T);
        a.out = f3.interior(); a.V = c3.D; a.H = c3.H; a.W = c3.W;
        a.out_row = (f3.W + 2 * f3.pad) * 8; a.out_plane = (f3.H + 2 * f3.pad) * a.out_row;
        a.tilesY = cdiv(c3.H, kH3TY); a.tilesX = cdiv(c3.W, kH3TXP); a.ntiles = c3.D * a.tilesY * a.tilesX;
        o.flops = 2.0 * c3.D * c3.H * c3.W * (9.0 * 8 * 8 + 4.0 * 32 * 8);  // (as the four launches count them: 9 taps of the composed layer, 2 x 2 half-resolution pixels per output pixel)
        o.bytes = 4.0 * (c3.n() + i2.n() + (double)c3.D * c3.H * c3.W * 8);
        ops_.push_back(o);
      } else {
      ConvLayer LA; LA.Cin = 8; LA.Cout = 8; LA.kd = 1; LA.kh = 3; LA.kw = 3; LA.weight = wa.data(); LA.bias = bint; LA.out_pad = fpad;
      emit_conv("fn.out3a", LA, CONV_XPAIR, c3, f3, nullptr, 0, 0);
      for (int py = 0; py < 2; ++py) {
        ConvLayer LB; LB.Cin = 32; LB.Cout = 8; LB.kd = 1; LB.kh = 3; LB.kw = 3; LB.weight = wo3.data.data(); LB.up2 = 1 + py; LB.out_pad = fpad;
        emit_conv(py ? "fn.out3c" : "fn.out3b", LB, CONV_NORMAL, i2, f3, f3.interior(), 1, f3.n() / 2);
      }
      Op o; o.kind = Op::BORDERFIX; o.name = "fn.out3d"; o.p2 = f3.interior(); o.p1 = plan_arena_->upload(T); o.d0 = c3.D; o.d1 = c3.H; o.d2 = c3.W;
      o.stage = fpad; o.bytes = 64.0 * c3.D * (c3.H + c3.W);
      ops_.push_back(o);
      }
    } else
    if (!sw_.no_skip_fusion && !sw_.skip_on_conv && kParityHooks && w3.dims[0] == 32 && w3.dims[1] == 8 && c3.C == 8 && i2.C == 32 &&
        i2.H * 2 == c3.H && i2.W * 2 == c3.W) {  // (parity build only: the fused-skip kernel instances are not in the product library)
      ConvFuse fz{c3.d, plan_arena_->upload(w3.data), plan_arena_->upload(blob_.at(fn + "skip.stage3.bias").data), i2.d, 8};
      DevTensor virt; virt.D = c3.D; virt.H = c3.H; virt.W = c3.W; virt.C = 32; virt.d = nullptr;  // inter3 exists in LDS only
      add_conv("fn.out3", fn + "out.stage3", "", false, false, virt, "feat3", 1, 3, 3, 1, 1, 1, false, CONV_XPAIR, nullptr, 0, &fz, fpad);
    } else {
      DevTensor &i3 = add_skip("fn.skip3", fn + "skip.stage3", c3, "inter3", i2);
      add_conv("fn.out3", fn + "out.stage3", "", false, false, i3, "feat3", 1, 3, 3, 1, 1, 1, false, CONV_XPAIR, nullptr, 0, nullptr, fpad);
    }
    fork_hi_ = ops_.size();
  }

  // ---------------------------------------------------------------- key-frame feature cache (drm_set_feature_cache)
  // In TANDEM six of a window's seven images were in the previous window (FullSystem.cpp:1162-1171 re-sends frameHessians[i]->image_bgr), and
  // FeatureNet is per image.  With the cache on, the engine keeps feat1..3 (and the u8 image) of the last `capacity` images; a window whose images
  // are all cached but at most ONE runs FeatureNet on that one view (its own single-view plan, the same kernel instances as the batch plan) and the
  // plane sweep reads every view's features where the cache holds them (CostVolArgs::vfeat): 15 launches over 7 views become 11 over one.
  //  * identity: a 128-bit key over a sample of the image (first / last 64 bytes + 512 evenly spaced 8-byte words) finds the entry; the hit is then
  //    made EXACT on the device -- every uploaded image is compared byte for byte with the entry's copy (k_verify_image), and a difference (a key
  //    collision) drops the whole cache and runs the window again without it, before any result leaves the engine;
  //  * windows with two or more uncached images (the first one, a reset) take the batch path and fill the cache from its outputs;
  //  * same bits with the cache on and off: the single-view plan is built from the batch plan's own choices (kernel family, channel pass, tiles per wave)
  //    -- the products and their order per output value do not depend on the tile shape or the batch size (tests: test_feature_cache_is_bit_identical);
  //  * OFF by default and in every leg of bench.py that feeds `value` / `single_window_ms` (they repeat one window: every image would hit).
  struct FnOut { size_t op; int stage; size_t offset; };  // launch `op` of the single-view plan stores into feat<stage + 1> at this float offset
  struct FcEntry { uint64_t key[2] = {0, 0}; float *feat[3] = {nullptr, nullptr, nullptr}; uint8_t *bgr = nullptr; uint64_t used = 0; bool valid = false; };
  static bool same_instance(const ConvLaunch &a, const ConvLaunch &b) {
    return a.async == b.async && a.ci == b.ci && a.ct == b.ct && a.pt == b.pt && a.fz == b.fz && a.bf3 == b.bf3 && a.nup == b.nup && a.ncw == b.ncw &&
           a.march.wino == b.march.wino && a.march.rm == b.march.rm;
  }
  void build_fn1(int H, int W) {
    fn1_ok_ = false; ops1_.clear(); fcache_.clear();
    if (ops_.empty() || ops_[0].kind != Op::FRONT || sw_.costvol_v1 || shard_nsrc_) return;  // (the single-view plan patches k_fn_front's image pointer per call)
    const size_t lo = fork_lo_, hi = fork_hi_, f2 = feat2_op_;
    std::vector<Op> batch;
    batch.swap(ops_);
    tprefix_ = "c1.";
    try { build_featurenet(1, H, W); } catch (...) { tprefix_.clear(); ops_.swap(batch); fork_lo_ = lo; fork_hi_ = hi; feat2_op_ = f2; throw; }
    tprefix_.clear();
    ops1_.swap(ops_); ops_.swap(batch);
    fork_lo_ = lo; fork_hi_ = hi; feat2_op_ = f2;
    if (!match_fn1()) { ops1_.clear(); return; }
    // the entries: three bordered feature maps + the image, per cached key frame
    const size_t img_bytes = (size_t)H * W * 3;
    fcache_.resize(fcache_cap_);
    for (FcEntry &e : fcache_) {
      for (int s = 0; s < 3; ++s) {
        const DevTensor &t = T("c1.feat" + std::to_string(s + 1));
        e.feat[s] = dalloc<float>(t.n()); misc_.push_back(e.feat[s]);
        DR_HIP(hipMemset(e.feat[s], 0, t.n() * 4));  // (the zero border is written once, here: the producers store interior pixels only)
      }
      e.bgr = dalloc<uint8_t>(img_bytes + 16); misc_.push_back(e.bgr);
    }
    if (!fc_flag_) { DR_HIP(hipHostMalloc((void **)&fc_flag_, sizeof(int), hipHostMallocDefault)); *fc_flag_ = 0; }
    if (!up_stream_) { DR_HIP(hipStreamCreateWithFlags(&up_stream_, hipStreamNonBlocking)); DR_HIP(hipEventCreateWithFlags(&ev_hits_, hipEventDisableTiming)); }
    // the single-view plan writes its three outputs straight into the entry of the image it runs on: which launches store into feat1..3, and where
    fn1_out_.clear();
    for (size_t i = 0; i < ops1_.size(); ++i) {
      Op &o = ops1_[i];
      float *out = o.kind == Op::CONV ? o.conv.args.out : (o.kind == Op::HEAD3 ? o.head3.out : nullptr);
      for (int s = 0; s < 3 && out; ++s) {
        const DevTensor &t = T("c1.feat" + std::to_string(s + 1));
        if (out >= t.d && out < t.d + t.n()) fn1_out_.push_back({i, s, (size_t)(out - t.d)});
      }
    }
    if (fn1_out_.size() != 3) { ops1_.clear(); fcache_.clear(); return; }
    fn1_ok_ = true;
  }
  // every launch of the single-view plan becomes the batch plan's own kernel instance for that layer (the planner ranks candidates by a cost model that sees
  // the batch size; the arithmetic of an instance does not)
  bool match_fn1() {
    for (Op &o : ops1_) {
      const Op *b = nullptr;
      for (size_t i = 0; i < fork_hi_ && i < ops_.size(); ++i) if (ops_[i].name == o.name) b = &ops_[i];
      if (!b || b->kind != o.kind) return false;
      if (o.kind == Op::FRONT || o.kind == Op::HEAD3) continue;
      if (o.kind != Op::CONV) return false;
      if (same_instance(o.conv, b->conv)) continue;
      if (!o.replan) return false;
      bool found = false;
      for (int r = 0; r < o.ncand && !found; ++r) {
        const ConvLaunch c = o.replan(r);
        if (same_instance(c, b->conv)) { o.conv = c; found = true; }
      }
      if (!found) return false;
    }
    return true;
  }
  static void image_key(const uint8_t *p, size_t n, int H, int W, uint64_t key[2]) {
    uint64_t a = 0xcbf29ce484222325ull ^ (uint64_t)H, b = 0x9e3779b97f4a7c15ull ^ (uint64_t)W;
    auto mix = [&](uint64_t w) { a = (a ^ w) * 0x100000001b3ull; b = (b + w) * 0xff51afd7ed558ccdull; b ^= b >> 29; };
    auto word = [&](size_t off) { uint64_t w; memcpy(&w, p + off, 8); return w; };
    for (size_t o = 0; o < 64; o += 8) { mix(word(o)); mix(word(n - 64 + o)); }
    const size_t step = (n / 512) & ~(size_t)7;
    for (size_t k = 1; k < 512 && step; ++k) mix(word(k * step));
    key[0] = a; key[1] = b;
  }
  // decides, for the window being staged, which views the cache answers.  fc_slot_[v]: the entry that holds (or will hold) view v's features.
  void plan_cache_use(int H, int W, int V, const uint8_t *const *bgrs, const std::vector<int> &order) {
    fc_fast_ = false; fc_fill_ = false; fc_miss_ = -1;
    if (!fn1_ok_ || (int)fcache_.size() < V + 1 || shard_nsrc_ || comm_ || phase_mode_) return;  // (a view-shard rank computes its own views every time)
    for (int s = 1; s <= 3; ++s) if (!cv5_applies(s)) return;  // (only k_costvol5 reads the views by pointer)
    const size_t img_bytes = (size_t)H * W * 3;
    uint64_t keys[8][2];
    int nmiss = 0;
    ++fc_clock_;
    for (int v = 0; v < V; ++v) {
      image_key(bgrs[order[v]], img_bytes, H, W, keys[v]);
      fc_slot_[v] = -1;
      for (size_t e = 0; e < fcache_.size(); ++e)
        if (fcache_[e].valid && fcache_[e].key[0] == keys[v][0] && fcache_[e].key[1] == keys[v][1]) { fc_slot_[v] = (int)e; break; }
      for (int u = 0; u < v; ++u) if (fc_slot_[v] >= 0 && fc_slot_[u] == fc_slot_[v]) fc_slot_[v] = -1;  // (two views with one key: only one may own the entry)
      if (fc_slot_[v] < 0) { ++nmiss; fc_miss_ = v; } else fcache_[fc_slot_[v]].used = fc_clock_;
    }
    auto evict = [&]() {  // the least recently used entry that this window does not use
      int best = -1;
      for (size_t e = 0; e < fcache_.size(); ++e) {
        bool in_window = false;
        for (int v = 0; v < V; ++v) in_window |= fc_slot_[v] == (int)e;
        if (!in_window && (best < 0 || !fcache_[e].valid || (fcache_[best].valid && fcache_[e].used < fcache_[best].used))) best = (int)e;
        if (best >= 0 && !fcache_[best].valid) break;
      }
      return best;
    };
    if (nmiss <= 1) {
      fc_fast_ = true;
      if (nmiss == 1) {
        const int e = evict();
        fcache_[e].valid = false;  // (valid again once a forward has enqueued its fill)
        fcache_[e].key[0] = keys[fc_miss_][0]; fcache_[e].key[1] = keys[fc_miss_][1]; fcache_[e].used = fc_clock_;
        fc_slot_[fc_miss_] = e;
      }
      fc_hits_ += V - nmiss; fc_misses_ += nmiss;
    } else {  // the batch path computes every view; its outputs fill the cache
      fc_fill_ = true; fc_miss_ = -1;
      for (int v = 0; v < V; ++v) {
        if (fc_slot_[v] >= 0) continue;
        const int e = evict();
        fcache_[e].valid = false;
        fcache_[e].key[0] = keys[v][0]; fcache_[e].key[1] = keys[v][1]; fcache_[e].used = fc_clock_;
        fc_slot_[v] = e;
      }
      fc_misses_ += V; ++fc_batch_windows_;
    }
    // where the plane sweep finds each view
    for (int s = 0; s < 3; ++s)
      for (int v = 0; v < V; ++v)
        if (fc_fast_) cv_[s].vfeat[v] = fcache_[fc_slot_[v]].feat[s];
  }
  void launch_fn_op(const Op &o, hipStream_t st) {
    if (o.kind == Op::CONV) launch_conv(o.conv, st);
    else if (o.kind == Op::FRONT) launch_fn_front(o.front, st);
    else if (o.kind == Op::HEAD3) launch_fn_head3(o.head3, st);
    else fail(DR_ERR_UNSUPPORTED, "feature cache: op kind %d in the single-view plan", (int)o.kind);
  }
  // one launch: the hits compared with their entries' images, the new image filed in its entry
  void launch_cache_io() {
    const size_t img_bytes = (size_t)H_ * W_ * 3;
    CacheIoArgs io{};
    for (int v = 0; v < V_; ++v) {
      io.img[v] = reinterpret_cast<const uint4 *>(d_bgr_ + v * img_bytes);
      io.entry[v] = reinterpret_cast<uint4 *>(fcache_[fc_slot_[v]].bgr);
    }
    io.miss = fc_miss_; io.flag = fc_flag_dev(); io.n16 = img_bytes / 16;
    hipLaunchKernelGGL(k_cache_io, dim3(32, V_), dim3(256), 0, stream_, io);
  }
  // fast path of a forward: verify the hits, FeatureNet on the one uncached view, its outputs (and image) into the entry
  void forward_cached_features() {
    const size_t img_bytes = (size_t)H_ * W_ * 3;
    if (!defer_cache_io_) launch_cache_io();
    if (fc_miss_ >= 0) {
      FcEntry &e = fcache_[fc_slot_[fc_miss_]];
      for (const FnOut &p : fn1_out_) {
        Op &o = ops1_[p.op];
        (o.kind == Op::CONV ? o.conv.args.out : o.head3.out) = e.feat[p.stage] + p.offset;
      }
      // (the heads on the side stream, as the batch path runs them, were measured: device time 1.821 -> 1.825 ms, the sliding loop x 1.10 instead of x 1.12-1.15 --
      // two cross-stream waits cost what the overlap of three small launches wins; not kept)
      for (Op &o : ops1_) {
        if (o.kind == Op::FRONT) o.front.bgr = d_bgr_ + fc_miss_ * img_bytes;
        launch_fn_op(o, stream_);
      }
      e.valid = true;
    }
  }
  // batch path with the cache on: every view's features (and image) into its entry, behind the forward that produced them
  void fill_cache_from_batch() {
    const size_t img_bytes = (size_t)H_ * W_ * 3;
    for (int v = 0; v < V_; ++v) {
      FcEntry &e = fcache_[fc_slot_[v]];
      if (e.valid) continue;
      for (int s = 0; s < 3; ++s) {
        const DevTensor &t = T("feat" + std::to_string(s + 1));
        const size_t n1 = t.n() / t.D;
        DR_HIP(hipMemcpyAsync(e.feat[s], t.d + v * n1, n1 * 4, hipMemcpyDeviceToDevice, stream_));
      }
      DR_HIP(hipMemcpyAsync(e.bgr, d_bgr_ + v * img_bytes, img_bytes, hipMemcpyDeviceToDevice, stream_));
      e.valid = true;
    }
  }
  int *fc_flag_dev() { int *d = nullptr; DR_HIP(hipHostGetDevicePointer((void **)&d, fc_flag_, 0)); return d; }
  // after a stream synchronise: a hit that was not one (key collision).  The cache is dropped and the staged window runs again on the batch path.
  bool cache_mismatch_recovered() {
    if (!fc_flag_ || !*fc_flag_) return false;
    *fc_flag_ = 0;
    ++fc_collisions_;
    for (FcEntry &e : fcache_) e.valid = false;
    fc_fast_ = false; fc_fill_ = false;
    for (int s = 0; s < 3; ++s) set_batch_vfeat(s);
    forward(nullptr);
    DR_HIP(hipStreamSynchronize(stream_));
    return true;
  }
  void set_batch_vfeat(int s) {
    const DevTensor &t = T("feat" + std::to_string(s + 1));
    for (int v = 0; v < V_ && v <= kMaxSrc; ++v) cv_[s].vfeat[v] = t.d + (size_t)v * (t.n() / t.D);
  }

  void build_plan(int H, int W, int V) {
    plan_arena_.reset(new DeviceArena());
    plan_arena_->err_flag = march_err_;
    for (float *&h : h_out_) if (h) { (void)hipHostFree(h); h = nullptr; }
    if (h_in_) { (void)hipHostFree(h_in_); h_in_ = nullptr; }
    for (int b = 0; b < 2; ++b) {
      DR_HIP(hipHostMalloc((void **)&h_out_[b], (size_t)H * W * 16, hipHostMallocDefault));
      DR_HIP(hipHostGetDevicePointer((void **)&h_out_dev_[b], h_out_[b], 0));
    }
    has_output_ = false;
    DR_HIP(hipHostMalloc((void **)&h_in_, (size_t)V * H * W * 3, hipHostMallocDefault));
    d_bgr_ = dalloc<uint8_t>((size_t)V * H * W * 3 + 16); misc_.push_back(d_bgr_);  // (+16: k_fn_front fetches a pixel as the two aligned words around it)
    d_state_ = dalloc<unsigned>(16); misc_.push_back(d_state_);   // four 4-word slots (the fused-scan form keeps one per level)
    d_hist_ = dalloc<unsigned>(3 * 2048); misc_.push_back(d_hist_);  // one histogram per level (the k_scan form uses the first only)
    DR_HIP(hipMemset(d_hist_, 0, 3 * 2048 * 4));
    DR_HIP(hipMemset(d_state_, 0, 16 * 4));

    build_featurenet(V, H, W);
    if (fcache_cap_ > 0) build_fn1(H, W);

    for (int s = 1; s <= 3; ++s) {
      const int sc = 1 << (3 - s), h = H / sc, w = W / sc, D = blob_.depth_num[s - 1], C = 32 >> (s - 1);
      if (!(D == 4 || D % 8 == 0)) fail(DR_ERR_UNSUPPORTED, "depth_num[%d]=%d must be 4 or a multiple of 8", s - 1, D);
      const std::string S = std::to_string(s), cr = "cost_regularization_net.stage" + S + ".", pre = "s" + S + ".";
      DevTensor &vol = alloc("volume" + S, D, h, w, C);
      if (C == 32 && sw_.vol_split && !conv_bf3_policy()) vol.split = 16;  // (the bf16 x 3 mode stages through its own kernel: one tensor)
      {
        Op o; o.kind = Op::COSTVOL; o.stage = s; o.name = pre + "costvol";
        o.bytes = 4.0 * ((double)V * h * w * C + vol.n());
        o.flops = (double)(V - 1) * D * h * w * (3.0 * C + 2.0 * C + 8.0 * C);  // diff^2, gate dot, weighted accumulate, 4-tap lerp
        ops_.push_back(o);
      }
      const bool four = D == 4;
      DevTensor &c0 = cbr3(pre + "conv0", cr + "conv0", vol, 1, 1, CONV_XPAIR);
      DevTensor &k1 = cbr3(pre + "conv1", cr + "conv1", c0, 2, 2, CONV_NORMAL);
      DevTensor &k2 = cbr3(pre + "conv2", cr + "conv2", k1, 1, 1, CONV_NORMAL);
      DevTensor &k3 = cbr3(pre + "conv3", cr + "conv3", k2, 2, 2, CONV_NORMAL);
      DevTensor &k4 = cbr3(pre + "conv4", cr + "conv4", k3, 1, 1, CONV_NORMAL);
      DevTensor &k5 = cbr3(pre + "conv5", cr + "conv5", k4, four ? 1 : 2, 2, CONV_NORMAL);
      DevTensor &k6 = cbr3(pre + "conv6", cr + "conv6", k5, 1, 1, CONV_NORMAL);
      DevTensor &x7 = dbr3(pre + "conv7", cr + "conv7", k6, four ? 1 : 2, k4);
      DevTensor &x9 = dbr3(pre + "conv9", cr + "conv9", x7, 2, k2);
#ifdef DR_PARITY_HOOKS
      const HostTensor &w11 = blob_.at(cr + "conv11.conv.weight");
      const HostTensor &wpr = blob_.at(cr + "prob.weight");
      const bool tail = sw_.tail_fused && !sw_.prob_on_conv && !conv_bf3_policy() && w11.dims.size() == 5 && w11.dims[0] == 16 && w11.dims[1] == 8 && w11.dims[2] == 3 &&
                        w11.dims[3] == 3 && w11.dims[4] == 3 && x9.C == 16 && c0.C == 8 && x9.D * 2 == D && x9.H * 2 == h && x9.W * 2 == w && wpr.dims[0] == 1 && wpr.dims[1] == 8;
      if (tail) {
        // conv11 (ConvTranspose3d 16 -> 8 + BN + ReLU, + conv0) and prob (8 -> 1) in one z-marching launch: the 8-channel full-resolution tensor
        // between them (78.6 MB at stages 2 and 3) is never written or read (tail_kernels.h)
        std::vector<float> sc, bi;
        fold_bn(cr + "conv11.bn", 8, sc, bi);
        sc.insert(sc.end(), bi.begin(), bi.end());
        DevTensor &lg = alloc("logits" + S, D, h, w, 1);
        Op o; o.kind = Op::TAIL; o.stage = s; o.name = pre + "tail";
        TailArgs &t = o.tail;
        t.x = x9.d; t.skip = c0.d; t.out = lg.d; t.D = D; t.h = h; t.w = w;
        t.wd = plan_arena_->upload(tail_pack_deconv(w11.data.data()));
        t.sb = plan_arena_->upload(sc);
        t.wp = plan_arena_->upload(tail_pack_prob(wpr.data.data()));
        const bool mf = sw_.tail_fused == 1;
        t.wmf = mf ? plan_arena_->upload(tail_pack_deconv_mfma(w11.data.data())) : nullptr;
        tail_pick_tile(h, w, t.QY, t.QX, mf);
        if (sw_.tail_qy == 4 || sw_.tail_qy == 8 || sw_.tail_qy == 16 || (sw_.tail_qy == 32 && !mf)) { t.QY = sw_.tail_qy; t.QX = 256 / t.QY; }
        t.zchunk = sw_.tail_zchunk > 0 ? std::min(D, sw_.tail_zchunk) : tail_pick_zchunk(D, h, w, t.QY, t.QX);
        const double N = (double)D * h * w;
        o.flops = 2.0 * 3.375 * 16 * 8 * N + 2.0 * 216 * N;  // the algorithmic MACs of both layers (halo recomputation not counted)
        o.bytes = 4.0 * (x9.n() + c0.n() + lg.n());
        ops_.push_back(o);
      } else
#endif
      {
      DevTensor &x11 = dbr3(pre + "conv11", cr + "conv11", x9, 2, c0);
      if (sw_.prob_on_conv && w % 8 == 0) {
        // A/B hook: the Cout = 1 head as 8 x-shifts per column on the MFMA kernel (CONV_X8, 50 % of the rows carry work);
        // measured against k_prob in profiles/r02_experiments.txt
        add_conv(pre + "prob", cr + "prob", "", false, false, x11, "logits" + S, 3, 3, 3, 1, 1, 1, false, CONV_X8, nullptr, 0);
      } else {
        const HostTensor &pw = blob_.at(cr + "prob.weight");  // (1,8,3,3,3) -> [tap][cin]
        std::vector<float> wt(27 * 8);
        for (int ci = 0; ci < 8; ++ci) for (int t = 0; t < 27; ++t) wt[t * 8 + ci] = pw.data[ci * 27 + t];
        DevTensor &lg = alloc("logits" + S, D, h, w, 1);
        Op o; o.kind = Op::PROB; o.stage = s; o.name = pre + "prob";
        o.p0 = x11.d; o.p1 = plan_arena_->upload(wt); o.p2 = lg.d; o.d0 = D; o.d1 = h; o.d2 = w;
        o.flops = 2.0 * 216 * D * h * w; o.bytes = 4.0 * (x11.n() + lg.n());
        ops_.push_back(o);
      }
      }
      alloc("depth" + S, 1, h, w, 1);
      alloc("conf" + S, 1, h, w, 1);
      { Op o; o.kind = Op::REGRESS; o.stage = s; o.name = pre + "regress"; o.bytes = 4.0 * ((double)D * h * w + 2.0 * h * w); o.flops = 8.0 * D * h * w; ops_.push_back(o); }
    }
    alloc("edge", 1, H, W, 1);
    alloc("depth", 1, H, W, 1);
    alloc("confidence", 1, H, W, 1);
    { Op o; o.kind = Op::EDGE; o.name = "filter.edge"; o.bytes = 8.0 * H * W; ops_.push_back(o); }
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int i = 0; i < 3; ++i) {
      Op o; o.kind = Op::HIST; o.shift = shifts[i]; o.bits = bits[i]; o.stage = i; o.d0 = i ? shifts[i - 1] : 0; o.d1 = i ? bits[i - 1] : 0;
      o.name = "filter.hist" + std::to_string(i); o.bytes = 4.0 * H * W; ops_.push_back(o);
      if (sw_.filter_fused) continue;  // (the scan of level i is the prologue of the next kernel: mvs_kernels.h k_hist_s / k_apply_s)
      Op q; q.kind = Op::SCAN; q.shift = shifts[i]; q.bits = bits[i]; q.name = "filter.scan" + std::to_string(i); ops_.push_back(q);
    }
    { Op o; o.kind = Op::APPLY; o.name = "filter.apply"; o.shift = shifts[2]; o.bits = bits[2]; o.bytes = 20.0 * H * W; ops_.push_back(o); }
  }

  // Copies the window into pinned memory in model order [ref, others] (dr_mvsnet.cpp:190-197), enqueues
  // the H2D copy and derives every per-call kernel parameter.  Caller holds mu_.
  void stage_inputs(int H, int W, int V, int ref, const uint8_t *const *bgrs, const float *K9, const float *const *c2ws,
                    float dmin, float dmax, float disc, bool may_prelaunch = false) {
    DR_HIP(hipSetDevice(device_));
    configure(H, W, V);
    const size_t img_bytes = (size_t)H * W * 3;
    std::vector<int> order;
    order.push_back(ref);
    for (int i = 0; i < V; ++i) if (i != ref) order.push_back(i);
    // view by view: the copy engine moves view v to the device while the host gathers view v + 1 into the pinned block (one 6.45 MB
    // transfer behind seven memcpys cost their sum: 0.3 + 0.2 ms at 640 x 480 x 7 on the operator boundary's critical path)
    // Images that already live in page-locked memory (drm_host_alloc, hipHostMalloc, hipHostRegister) go to the device straight from
    // where they are -- no gather into the pinned block, which is the 0.3-0.4 ms memcpy on the operator boundary's critical path; the
    // call still returns only when the copies have completed ("inputs are copied before return": the caller may reuse its buffers).
    // "Page-locked" is decided for the WHOLE image, not for its first byte: the allocation (or registered range) the first byte lies in has
    // to contain all img_bytes of it -- an image that starts inside a hipHostRegister'ed range and runs past its end takes the staging path.
    bool pinned = true;
    for (int v = 0; v < V && pinned; ++v) {
      hipPointerAttribute_t at;
      hipDeviceptr_t base = nullptr;
      size_t span = 0;
      if (hipPointerGetAttributes(&at, bgrs[v]) != hipSuccess || at.type != hipMemoryTypeHost || !at.devicePointer ||
          hipMemGetAddressRange(&base, &span, (hipDeviceptr_t)at.devicePointer) != hipSuccess) {
        (void)hipGetLastError();
        pinned = false;
      } else {
        const char *lo = (const char *)base, *p = (const char *)at.devicePointer;
        pinned = p >= lo && p + img_bytes <= lo + span;
      }
    }
    // stage intrinsics: rows 0-1 x 0.25 / 0.5 / 1 (the C++ rule, dr_mvsnet.cpp:226-247)
    double w2c[8][16];
    for (int v = 0; v < V; ++v) {
      double c2w[16];
      for (int i = 0; i < 16; ++i) c2w[i] = c2ws[order[v]][i];
      inv4(c2w, w2c[v]);
    }
    const float base_interval = (dmax - dmin) / (float)(blob_.depth_num[0] - 1);  // module.py:1493
    for (int s = 1; s <= 3; ++s) {
      const float f = s == 1 ? 0.25f : (s == 2 ? 0.5f : 1.f);
      float Ks[9];
      for (int i = 0; i < 9; ++i) Ks[i] = i < 6 ? (float)((double)f * (double)K9[i]) : K9[i];
      const int sc = 1 << (3 - s), h = H / sc, w = W / sc, D = blob_.depth_num[s - 1], C = 32 >> (s - 1);
      CostVolArgs &a = cv_[s - 1];
      memset(&a, 0, sizeof a);
      a.feat = T("feat" + std::to_string(s)).d;
      a.fpad = T("feat" + std::to_string(s)).pad;
      a.vol = T("volume" + std::to_string(s)).d;
      a.split = T("volume" + std::to_string(s)).split;
      a.V = V; a.h = h; a.w = w;
      a.dchunk = s == 1 ? 4 : (D >= 16 ? 8 : D);  // enough workgroups to fill 256 CUs at every stage
      a.view_aggregation = blob_.view_aggregation;
      // k_costvol5 holds a chunk's planes in registers: 4 planes leave room for six waves per SIMD (0.150 -> 0.131 ms at stage 2, 0.117 -> 0.106 at stage 3)
      if (a.view_aggregation && a.fpad && D % 4 == 0 && !sw_.costvol_v2 && !sw_.costvol_v3 && !sw_.cv4_stages) a.dchunk = 4;
      if (sw_.cv_dchunk[s - 1] > 0) a.dchunk = std::min(D, sw_.cv_dchunk[s - 1]);  // tuning hook
      // view sharding: this rank's window holds a subset of the source views, the divisor stays the whole window's
      if (shard_nsrc_ && !blob_.view_aggregation) fail(DR_ERR_UNSUPPORTED, "view sharding needs a view-aggregation model (the variance volume is not a sum over views)");
      a.nsrc_f = shard_nsrc_ ? (float)shard_nsrc_ : (float)(V - 1);
      PlaneArgs &p = a.planes;
      p.D = D; p.dmin = dmin; p.interval = base_interval;
      if (s > 1) {
        p.prev = T("depth" + std::to_string(s - 1)).d; p.hp = h / 2; p.wp = w / 2;
        const float delta = blob_.ratio[s - 1] * base_interval;  // cva_mvsnet.py:151
        p.half_range = ((float)D / 2.f) * delta;                  // module.py:1518
        p.full_range = (float)D * delta;                          // module.py:1526
      }
      double r_w2p[16], r_p2w[16];
      world_to_pixel(Ks, w2c[0], r_w2p);
      inv4(r_w2p, r_p2w);
      for (int v = 1; v < V; ++v) {
        double s_w2p[16], M[16];
        world_to_pixel(Ks, w2c[v], s_w2p);
        mul4(s_w2p, r_p2w, M);
        for (int i = 0; i < 12; ++i) a.M[v - 1][i] = (float)M[i];
      }
      if (blob_.view_aggregation) {
        const std::string g = "volume_gates.stage" + std::to_string(s) + ".";
        const auto &w0 = blob_.at(g + "0.weight").data;
        for (int c = 0; c < C; ++c) a.gw[c] = w0[c];
        auto bnf = [&](const std::string &bn, double &A, double &B) {
          const double ga = blob_.at(bn + ".weight").data[0], be = blob_.at(bn + ".bias").data[0];
          const double mu = blob_.at(bn + ".running_mean").data[0], var = blob_.at(bn + ".running_var").data[0];
          A = ga / std::sqrt(var + 1e-5); B = be - mu * A;
        };
        double A1, B1, A2, B2;
        bnf(g + "1", A1, B1); bnf(g + "4", A2, B2);
        const double b0 = blob_.at(g + "0.bias").data[0], w3 = blob_.at(g + "3.weight").data[0], b3 = blob_.at(g + "3.bias").data[0];
        a.gA1 = (float)A1; a.gB1 = (float)(b0 * A1 + B1);
        a.gA2 = (float)(w3 * A2); a.gB2 = (float)(b3 * A2 + B2);
      }
      RegressArgs &r = rg_[s - 1];
      r.logits = T("logits" + std::to_string(s)).d;
      r.depth = T("depth" + std::to_string(s)).d;
      r.conf = T("conf" + std::to_string(s)).d;
      r.planes = p; r.h = h; r.w = w;
    }
    for (int s = 0; s < 3; ++s) set_batch_vfeat(s);
    if (fcache_cap_ > 0) plan_cache_use(H, W, V, bgrs, order);
    // quantile rank, computed in float32 like module.py:1348-1349
    const float hw = (float)(H * W);
    float cut = hw * (100.f - disc);
    cut = cut / 100.f;
    long long ci = (long long)cut;
    if (ci < 0) ci = 0;
    if (ci > (long long)H * W - 1) ci = (long long)H * W - 1;
    filter_rank_ = (unsigned)ci;
    prelaunched_ = false;
    // PRELAUNCH (CallAsync with the feature cache answering the window): the device needs ONE image -- the new one -- to start; the other six are only compared
    // with their cache entries, which can happen last.  So the new image goes up first, the helper thread stages and uploads the rest on a second stream, and
    // THIS thread enqueues the whole forward meanwhile; the comparison (k_cache_io) is enqueued behind it once the uploads are in flight.  The device starts
    // ~0.3 ms earlier in TandemBackend's loop (it idles while a window is staged); the call still returns only when every image has been copied.
    if (may_prelaunch && fc_fast_ && up_stream_) {
      auto up_one = [&, pinned](int v, hipStream_t st) {
        const uint8_t *src = bgrs[order[v]];
        if (!pinned) { memcpy(h_in_ + v * img_bytes, src, img_bytes); src = h_in_ + v * img_bytes; }
        DR_HIP(hipMemcpyAsync(d_bgr_ + v * img_bytes, src, img_bytes, hipMemcpyHostToDevice, st));
      };
      const int miss = fc_miss_;
      if (miss >= 0) {
        up_one(miss, stream_);
        if (pinned) {  // (read in place from the caller's page-locked image: waited for before the call returns)
          if (!ev_h2d_) DR_HIP(hipEventCreateWithFlags(&ev_h2d_, hipEventDisableTiming));
          DR_HIP(hipEventRecord(ev_h2d_, stream_));
        }
      }
      copier_.run([&, miss] {
        DR_HIP(hipSetDevice(device_));
        for (int v = 0; v < V; ++v) if (v != miss) up_one(v, up_stream_);
        DR_HIP(hipEventRecord(ev_hits_, up_stream_));
      });
      struct Count {  // this window counts as in flight from here; the worker's InFlight takes the count over once prelaunched_ is set
        std::atomic<int> &n; bool handed_over = false;
        explicit Count(std::atomic<int> &a) : n(a) { n.fetch_add(1, std::memory_order_relaxed); }
        ~Count() { if (!handed_over) n.fetch_sub(1, std::memory_order_relaxed); }
      } count(windows_in_flight(device_));
      try {
        defer_cache_io_ = true;
        forward(nullptr);
        defer_cache_io_ = false;
      } catch (...) { defer_cache_io_ = false; copier_.wait_quiet(); throw; }
      copier_.wait();
      DR_HIP(hipStreamWaitEvent(stream_, ev_hits_, 0));
      launch_cache_io();
      if (pinned) {  // uploaded in place from the caller's page-locked images: they must have been read before the call returns
        DR_HIP(hipEventSynchronize(ev_hits_));
        if (miss >= 0) DR_HIP(hipEventSynchronize(ev_h2d_));
      }
      prelaunched_ = true;
      count.handed_over = true;
    } else {
    auto upload_views = [&, pinned](int v0) {  // views v0, v0 + 2, ...: gather into the staging block (unless page-locked already), then the copy engine
        DR_HIP(hipSetDevice(device_));
        for (int v = v0; v < V; v += 2) {
          const uint8_t *src = bgrs[order[v]];
          if (!pinned) { memcpy(h_in_ + v * img_bytes, src, img_bytes); src = h_in_ + v * img_bytes; }
          DR_HIP(hipMemcpyAsync(d_bgr_ + v * img_bytes, src, img_bytes, hipMemcpyHostToDevice, stream_));
        }
      };
      if (!pinned) {
        copier_.run([&] { upload_views(1); });  // the helper thread takes every other view
        try { upload_views(0); } catch (...) { copier_.wait_quiet(); throw; }  // (the job refers to this frame's locals)
        copier_.wait();
      } else {
        upload_views(1);
        upload_views(0);
      }
      if (pinned) {
        if (!ev_h2d_) DR_HIP(hipEventCreateWithFlags(&ev_h2d_, hipEventDisableTiming));
        DR_HIP(hipEventRecord(ev_h2d_, stream_));
        DR_HIP(hipEventSynchronize(ev_h2d_));
      }

    }
  }

  // Enqueue one complete forward on stream_ (or the ops [first, last) of it).  ev (optional): ops_.size()+1 events
  // for per-op timing.
  //
  // A whole forward (no per-op events, no range) forks: the FeatureNet heads that only stages 2 and 3 need
  // (fn.skip2, fn.out2, fn.skip3, fn.out3 -- 0.5 ms at 640x480) run on a side stream under stage 1's cost volume and
  // regularisation, whose coarse UNet levels leave most CUs idle; stage 2 / 3's cost volume waits for feat2 / feat3.
  // The next forward's main-stream work is ordered after those waits, so the side stream never runs ahead of a reader.
  void forward(std::vector<hipEvent_t> *ev, size_t first = 0, size_t last = ~(size_t)0) {
    if ((fc_fast_ || fc_fill_) && !(first == 0 && last >= ops_.size())) {  // a partial forward (phases, one op) cannot skip FeatureNet: the window goes back to the batch path
      fc_fast_ = fc_fill_ = false;
      for (int s = 0; s < 3; ++s) set_batch_vfeat(s);
    }
    const bool cached = fc_fast_;  // FeatureNet answered by the feature cache: its ops are skipped
    // ... unless other engines of this process have windows in flight on the device: their kernels already fill the idle CUs, and a second stream per
    // engine only adds queue contention (4 engines: 580 depth maps/s without the fork against 570 with it, profiles/r06_queues_side_stream.txt; alone: 2.150 ms
    // with it against 2.165).  Same kernels in the same per-stream order either way: the result does not depend on it.
    const bool alone = windows_in_flight(device_).load(std::memory_order_relaxed) <= 1;
    const bool fork = side_enabled_ && alone && !ev && first == 0 && last >= ops_.size() && fork_lo_ < fork_hi_ && !cached;
    if (cached) forward_cached_features();
    // view shard, reduce-to-root form: between a stage's cost volume and its regression only rank 0 works
    const bool rooted = comm_ && shard_nsrc_ && !phase_mode_ && !sw_.shard_allreduce;
    bool idle_stage = false;
    for (bool &f : regress_done_) f = false;  // (a PROB op and the REGRESS op it may answer for always lie in the same call; nothing is carried over from a call that threw)
    size_t i = 0;
    for (const Op &o : ops_) {
      if (ev) DR_HIP(hipEventRecord((*ev)[i], stream_));
      ++i;
      if (i - 1 < first || i - 1 >= last) continue;
      if (cached && i - 1 < fork_hi_) continue;
      if (idle_stage && o.kind != Op::REGRESS) continue;  // (CostRegNet and prob of this stage run on rank 0 only)
      const bool on_side = fork && i - 1 >= fork_lo_ && i - 1 < fork_hi_;
      if (fork && i - 1 == fork_lo_) { DR_HIP(hipEventRecord(ev_fork_, stream_)); DR_HIP(hipStreamWaitEvent(side_, ev_fork_, 0)); }
      if (fork && o.kind == Op::COSTVOL && o.stage == 2) DR_HIP(hipStreamWaitEvent(stream_, ev_feat2_, 0));
      if (fork && o.kind == Op::COSTVOL && o.stage == 3) DR_HIP(hipStreamWaitEvent(stream_, ev_feat3_, 0));
      hipStream_t stream_ = on_side ? side_ : this->stream_;  // shadows the member for the launches below
      switch (o.kind) {
        case Op::PREPROCESS: {
          const size_t npix = (size_t)V_ * H_ * W_;
          hipLaunchKernelGGL(k_preprocess, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream_, d_bgr_,
                             reinterpret_cast<float4 *>(T("image").d), lut_, npix);
          break;
        }
        case Op::CONV:
          launch_conv(o.conv, stream_);
          break;
        case Op::BORDERFIX: {
          const int per = 2 * (o.d1 + o.d2) - 4, n = o.d0 * per * 8;
          const int rs = (o.d2 + 2 * o.stage) * 8;
          hipLaunchKernelGGL(k_out3_border, dim3(cdiv(n, 256)), dim3(256), 0, stream_, o.p2, o.p1, o.d0, o.d1, o.d2, rs, (size_t)(o.d1 + 2 * o.stage) * rs);
          break;
        }
        case Op::SKIPUP: {
#ifdef DR_PARITY_HOOKS
          const size_t npix = (size_t)o.d0 * o.d1 * o.d2;
          const dim3 grid((unsigned)std::min<size_t>((npix + 31) / 32, 8192));
          hipLaunchKernelGGL(k_skip_up<8>, grid, dim3(256), 0, stream_, o.p0, o.p1, o.p3, o.p4, o.p2, o.d0, o.d1, o.d2);
#endif
          break;
        }
        case Op::PROB: {
          // z-march chunk: long chunks amortise the 2 halo planes, but the launch needs ~1000 waves to fill the chip
          // (round-2 sweep: 48x120x160 -> 4, 32x240x320 -> 8, 8x480x640 -> 8)
#ifdef DR_PARITY_HOOKS
          if (sw_.prob_v1) {  // round 2's L1-gather kernel: one output column per lane (r2 sweep: 4x the waves beats the 4-column variant)
            int zchunk = std::min(o.d0, 8);
            while (zchunk > 2 && cdiv(o.d1 * (o.d2 / 4), 64) * cdiv(o.d0, zchunk) < 800) zchunk /= 2;
            if (sw_.prob_zchunk > 0) zchunk = std::min(o.d0, sw_.prob_zchunk);
            const int pb = sw_.prob_block, xo = sw_.prob_xo == 2 ? 2 : (sw_.prob_xo == 4 ? 4 : 1);
            dim3 grid(cdiv(o.d1 * (o.d2 / xo), pb), cdiv(o.d0, zchunk));
            int gz = 0, nwg = 0;
            if (!sw_.prob_launch_order) {  // XCD-band workgroup order (A/B hook: the plain 2-D launch order)
              gz = (int)grid.y; nwg = (int)(grid.x * grid.y);
              grid = dim3(8 * cdiv(nwg, 8));
            }
            if (xo == 4) hipLaunchKernelGGL(k_prob<4>, grid, dim3(pb), 0, stream_, o.p0, o.p1, o.p2, o.d0, o.d1, o.d2, zchunk, gz, nwg);
            else if (xo == 2) hipLaunchKernelGGL(k_prob<2>, grid, dim3(pb), 0, stream_, o.p0, o.p1, o.p2, o.d0, o.d1, o.d2, zchunk, gz, nwg);
            else hipLaunchKernelGGL(k_prob<1>, grid, dim3(pb), 0, stream_, o.p0, o.p1, o.p2, o.d0, o.d1, o.d2, zchunk, gz, nwg);
            break;
          }
#endif
          // LDS-staged plane tiles (k_prob2<NR>: NR rows per lane, tile 4 NR x 64)
          const int NR = sw_.prob_rows == 2 || sw_.prob_rows == 4 ? sw_.prob_rows : 1;  // (measured: 0.028 / 0.044 / 0.084 ms at stage 2 for 1 / 2 / 4 rows per lane -- fewer, fatter workgroups lose more than the shared reads win)
          const int tyr = kProbTY * NR;
          int zc = std::min(o.d0, 8);
          while (zc > 2 && cdiv(o.d1, tyr) * cdiv(o.d2, kProbTX) * cdiv(o.d0, zc) < (NR == 1 ? 1024 : 512)) zc /= 2;  // enough workgroups for every CU's LDS
          if (sw_.prob_zchunk > 0) zc = std::min(o.d0, sw_.prob_zchunk);
          const int gxp = cdiv(o.d2, kProbTX), gyp = cdiv(o.d1, tyr), gzp = cdiv(o.d0, zc), nw = gxp * gyp * gzp;
          const size_t pl = prob2_lds_bytes(NR);
          if (NR == 4) {
            static std::atomic<int> big{0};
            if (!big.load()) { DR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_prob2<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); big.store(1); }
            hipLaunchKernelGGL(k_prob2<4>, dim3(8 * cdiv(nw, 8)), dim3(256), pl, stream_, o.p0, o.p1, o.p2, o.d0, o.d1, o.d2, zc, gxp, gyp, gzp, nw);
          } else if (NR == 2) hipLaunchKernelGGL(k_prob2<2>, dim3(8 * cdiv(nw, 8)), dim3(256), pl, stream_, o.p0, o.p1, o.p2, o.d0, o.d1, o.d2, zc, gxp, gyp, gzp, nw);
          else if (sw_.prob_regress && gzp == 1 && o.d0 == 8 && o.stage >= 1 && o.stage <= 3 && !sw_.regress_generic && !idle_stage) {
            // all planes are one depth chunk: the regression follows in the lane that produced the logits (k_prob2_regress); the REGRESS op of this stage then has nothing to launch
            hipLaunchKernelGGL(k_prob2_regress<8>, dim3(8 * cdiv(nw, 8)), dim3(256), pl, stream_, o.p0, o.p1, o.p2, o.d1, o.d2, gxp, gyp, nw, rg_[o.stage - 1]);
            regress_done_[o.stage - 1] = true; prob_fused_last_[o.stage - 1] = true;
          } else {
            hipLaunchKernelGGL(k_prob2<1>, dim3(8 * cdiv(nw, 8)), dim3(256), pl, stream_, o.p0, o.p1, o.p2, o.d0, o.d1, o.d2, zc, gxp, gyp, gzp, nw);
            if (o.stage >= 1 && o.stage <= 3) prob_fused_last_[o.stage - 1] = false;
          }
          break;
        }
        case Op::TAIL:
#ifdef DR_PARITY_HOOKS
          launch_tail(o.tail, stream_);
#endif
          break;
        case Op::FRONT:
          launch_fn_front(o.front, stream_);
          break;
        case Op::HEAD3:
          launch_fn_head3(o.head3, stream_);
          break;
        case Op::COSTVOL: {
          const CostVolArgs &a = cv_[o.stage - 1];
          const int C = 32 >> (o.stage - 1);
          if (a.V - 1 <= 0) {  // a view-shard rank that holds the reference view only: its partial volume is the empty sum (the kernels return without storing)
            const DevTensor &vol0 = T("volume" + std::to_string(o.stage));
            DR_HIP(hipMemsetAsync(vol0.d, 0, vol0.n() * 4, stream_));
          }
          CostVolArgs b = a;
          b.gz = cdiv(a.planes.D, a.dchunk);
          b.abl = sw_.cv5_abl;
#ifdef DR_PARITY_HOOKS
          if (!a.fpad) {  // DR_COSTVOL_V1: round 2's kernel on unpadded feature maps; channels per lane 4 (fewest L1 line accesses per byte) or 8
            const int cpl = C >= 16 ? sw_.costvol_cpl : 4, pxb = 256 / (C / cpl);
            b.gx = cdiv(a.w, pxb); b.nwg = b.gx * b.gz * a.h;
            const dim3 grid1(8 * cdiv(b.nwg, 8));
            if (C == 32 && cpl == 8) hipLaunchKernelGGL((k_costvol<32, 8>), grid1, dim3(256), 0, stream_, b);
            else if (C == 32) hipLaunchKernelGGL((k_costvol<32, 4>), grid1, dim3(256), 0, stream_, b);
            else if (C == 16 && cpl == 8) hipLaunchKernelGGL((k_costvol<16, 8>), grid1, dim3(256), 0, stream_, b);
            else if (C == 16) hipLaunchKernelGGL((k_costvol<16, 4>), grid1, dim3(256), 0, stream_, b);
            else hipLaunchKernelGGL((k_costvol<8, 4>), grid1, dim3(256), 0, stream_, b);
          } else
#endif
          {  // bordered feature maps: 4 channels per lane, no per-tap validity logic
            b.gx = cdiv(a.w, 1024 / C); b.nwg = b.gx * b.gz * a.h;
            const dim3 grid(8 * cdiv(b.nwg, 8));
#ifdef DR_PARITY_HOOKS  // k_costvol4 (source taps staged through LDS): view-aggregation models, whole pixel tiles, depth chunks of 8 (4 when D = 4)
            const int dch = a.planes.D >= 8 ? 8 : 4;
            const int tw = C == 8 ? 16 : 8, th = (1024 / C) / tw;
            if (cv4_applies(o.stage)) {
              CostVolArgs c4 = a;
              c4.gx = a.w / tw; c4.gz = a.planes.D / dch; c4.nwg = c4.gx * (a.h / th) * c4.gz;
              const dim3 g4(8 * cdiv(c4.nwg, 8));
              // planes per step: 8 where neighbouring planes move a sample by a fraction of a pixel (the box hardly grows), else 4
              const bool sp8 = dch == 8 && C != 32 && ((sw_.cv4_sp8 >> (o.stage - 1)) & 1);
              if (C == 32 && dch == 8) hipLaunchKernelGGL((k_costvol4<32, 8, 4>), g4, dim3(256), 0, stream_, c4);
              else if (C == 32) hipLaunchKernelGGL((k_costvol4<32, 4, 4>), g4, dim3(256), 0, stream_, c4);
              else if (C == 16 && sp8) hipLaunchKernelGGL((k_costvol4<16, 8, 8>), g4, dim3(256), 0, stream_, c4);
              else if (C == 16 && dch == 8) hipLaunchKernelGGL((k_costvol4<16, 8, 4>), g4, dim3(256), 0, stream_, c4);
              else if (C == 16) hipLaunchKernelGGL((k_costvol4<16, 4, 4>), g4, dim3(256), 0, stream_, c4);
              else if (sp8) hipLaunchKernelGGL((k_costvol4<8, 8, 8>), g4, dim3(256), 0, stream_, c4);
              else if (dch == 8) hipLaunchKernelGGL((k_costvol4<8, 8, 4>), g4, dim3(256), 0, stream_, c4);
              else hipLaunchKernelGGL((k_costvol4<8, 4, 4>), g4, dim3(256), 0, stream_, c4);
            } else
#endif
            if (cv5_applies(o.stage)) {  // view-outer / plane-inner sweep, the chunk's planes accumulate in registers (bit-identical to k_costvol3)
              const bool d8 = a.dchunk == 8;
              CostVolArgs b4 = b;  // the four-row tile: x segments of a quarter of the pixels, four rows per workgroup
              b4.gx = cdiv(a.w, 256 / C); b4.nwg = b4.gx * b4.gz * cdiv(a.h, 4);
              const dim3 grid4(8 * cdiv(b4.nwg, 8));
              [[maybe_unused]] const bool rows4 = sw_.cv5_rows ? sw_.cv5_rows == 4 : true;  // (four-row tiles at every stage since the single-set form: 0.126 -> 0.122 ms at stage 2, stage 1 unchanged)
#ifdef DR_PARITY_HOOKS
#define DR_CV5(CC, DD) do { if (!sw_.cv5_reuse) hipLaunchKernelGGL((k_costvol5<CC, DD, 0, 1>), grid, dim3(256), 0, stream_, b); \
                            else if (rows4) hipLaunchKernelGGL((k_costvol5<CC, DD, 1, 4>), grid4, dim3(256), 0, stream_, b4); \
                            else hipLaunchKernelGGL((k_costvol5<CC, DD, 1, 1>), grid, dim3(256), 0, stream_, b); } while (0)
#else
#define DR_CV5(CC, DD) hipLaunchKernelGGL((k_costvol5<CC, DD, 1, 4>), grid4, dim3(256), 0, stream_, b4)
#endif
              if (C == 32 && d8) DR_CV5(32, 8);
              else if (C == 32) DR_CV5(32, 4);
              else if (C == 16 && d8) DR_CV5(16, 8);
              else if (C == 16) DR_CV5(16, 4);
              else if (d8) DR_CV5(8, 8);
              else DR_CV5(8, 4);
#undef DR_CV5
            } else {
            // k_costvol3 (the lanes of a pixel share the per-sample set-up) needs whole batches of 4 iterations per depth chunk
            const bool v3 = !sw_.costvol_v2 && a.dchunk % 4 == 0 && a.planes.D % 4 == 0;
            if (v3 && C == 32) hipLaunchKernelGGL((k_costvol3<32>), grid, dim3(256), 0, stream_, b);
            else if (v3 && C == 16) hipLaunchKernelGGL((k_costvol3<16>), grid, dim3(256), 0, stream_, b);
            else if (v3) hipLaunchKernelGGL((k_costvol3<8>), grid, dim3(256), 0, stream_, b);
            else if (C == 32) hipLaunchKernelGGL((k_costvol2<32>), grid, dim3(256), 0, stream_, b);
            else if (C == 16) hipLaunchKernelGGL((k_costvol2<16>), grid, dim3(256), 0, stream_, b);
            else hipLaunchKernelGGL((k_costvol2<8>), grid, dim3(256), 0, stream_, b);
            }
          }
          if (comm_ && shard_nsrc_ && !phase_mode_) {  // view shard: sum the partial volumes of all ranks, in place, in stream order
            const DevTensor &vol = T("volume" + std::to_string(o.stage));
            Rccl &r = Rccl::get();
            if (rooted) {
              r.check(r.Reduce(vol.d, vol.d, vol.n(), ncclFloat, ncclSum, 0, comm_, stream_), "ncclReduce");
              idle_stage = comm_rank_ != 0;
            } else r.check(r.AllReduce(vol.d, vol.d, vol.n(), ncclFloat, ncclSum, comm_, stream_), "ncclAllReduce");
          }
          break;
        }
        case Op::REGRESS: {
          const RegressArgs &r = rg_[o.stage - 1];
          const bool done = regress_done_[o.stage - 1];  // (k_prob2_regress did it)
          regress_done_[o.stage - 1] = false;
          if (!idle_stage && !done) {
            const dim3 grid(cdiv(r.h * r.w, 256)), block(256);
            const int D = sw_.regress_generic ? 0 : r.planes.D;  // DR_REGRESS_GENERIC=1: the three-pass kernel for every plane count (A/B and parity hook)
            if (D == 48) hipLaunchKernelGGL(k_regress_r<48>, grid, block, 0, stream_, r);
            else if (D == 32) hipLaunchKernelGGL(k_regress_r<32>, grid, block, 0, stream_, r);
            else if (D == 8) hipLaunchKernelGGL(k_regress_r<8>, grid, block, 0, stream_, r);
            else if (D == 4) hipLaunchKernelGGL(k_regress_r<4>, grid, block, 0, stream_, r);
            else hipLaunchKernelGGL(k_regress, grid, block, 0, stream_, r);
          }
          if (rooted) {  // the stage's depth map goes back to every rank: stage s + 1 centres its hypotheses on it; stage 3's
            Rccl &c = Rccl::get();  // depth and confidence are the result (the edge filter then runs on every rank: 0.08 ms)
            const DevTensor &dep = T("depth" + std::to_string(o.stage));
            c.check(c.Broadcast(dep.d, dep.d, dep.n(), ncclFloat, 0, comm_, stream_), "ncclBroadcast");
            if (o.stage == 3) {
              const DevTensor &cf = T("conf3");
              c.check(c.Broadcast(cf.d, cf.d, cf.n(), ncclFloat, 0, comm_, stream_), "ncclBroadcast");
            }
            idle_stage = false;
          }
          break;
        }
        case Op::EDGE:
          if (sw_.filter_fused) hipLaunchKernelGGL(k_edge2, dim3(cdiv(H_ * W_, 256)), dim3(256), 0, stream_, T("depth3").d, T("edge").d, H_, W_, d_state_, d_hist_, filter_rank_);
          else hipLaunchKernelGGL(k_edge, dim3(cdiv(H_ * W_, 256)), dim3(256), 0, stream_, T("depth3").d, T("edge").d, H_, W_, d_state_, filter_rank_);
          break;
        case Op::HIST:
          if (sw_.filter_fused && o.stage > 0)
            hipLaunchKernelGGL(k_hist_s, dim3(std::min(cdiv(H_ * W_, 256), sw_.hist_blocks)), dim3(256), 0, stream_, T("edge").d, H_ * W_, o.d0, o.d1, o.shift, o.bits, o.stage,
                               d_state_, d_hist_);
          else hipLaunchKernelGGL(k_hist, dim3(std::min(cdiv(H_ * W_, 256), sw_.hist_blocks)), dim3(256), 0, stream_, T("edge").d, H_ * W_, o.shift, o.bits, d_state_, d_hist_);
          break;
        case Op::SCAN:
          hipLaunchKernelGGL(k_scan, dim3(1), dim3(256), 0, stream_, d_state_, d_hist_, o.shift, o.bits);
          break;
        case Op::APPLY:
          if (sw_.filter_fused)
            hipLaunchKernelGGL(k_apply_s, dim3(cdiv(H_ * W_, 256)), dim3(256), 0, stream_, T("edge").d, d_state_, d_hist_, o.shift, o.bits, T("depth3").d, T("conf3").d,
                               T("depth").d, T("confidence").d, H_ * W_);
          else hipLaunchKernelGGL(k_apply, dim3(cdiv(H_ * W_, 256)), dim3(256), 0, stream_, T("edge").d, d_state_, T("depth3").d, T("conf3").d,
                                  T("depth").d, T("confidence").d, H_ * W_);
          break;
      }
      // the side stream's results are published after whichever op ends them (fn.out2; the LAST op of the fork range: a
      // convolution in the fused-skip form, the border kernel in the folded form) -- independent of the op kind
      if (on_side && i - 1 == feat2_op_) DR_HIP(hipEventRecord(ev_feat2_, side_));
      if (on_side && i - 1 == fork_hi_ - 1) DR_HIP(hipEventRecord(ev_feat3_, side_));
    }
    if (fc_fill_ && first == 0 && last >= ops_.size()) fill_cache_from_batch();
    if (ev) DR_HIP(hipEventRecord((*ev)[i], stream_));
    DR_HIP(hipGetLastError());
  }

  // k_costvol5 (view-outer / plane-inner sweep): view-aggregation models, bordered feature maps, depth chunks of exactly 4 or 8 planes
  bool cv5_applies(int stage) const {
    const CostVolArgs &a = cv_[stage - 1];
    return !sw_.costvol_v1 && !sw_.costvol_v2 && !sw_.costvol_v3 && a.fpad && a.view_aggregation && a.V > 1 && (a.dchunk == 4 || a.dchunk == 8) &&
           a.planes.D % a.dchunk == 0;
  }
  // k_costvol4 (taps staged through LDS): view-aggregation models, bordered feature maps, whole pixel tiles, whole depth chunks
  bool cv4_applies(int stage) const {
    const CostVolArgs &a = cv_[stage - 1];
    const int C = 32 >> (stage - 1), tw = C == 8 ? 16 : 8, th = (1024 / C) / tw, dch = a.planes.D >= 8 ? 8 : 4;
    return !sw_.costvol_v1 && !sw_.costvol_v2 && !sw_.costvol_v3 && ((sw_.cv4_stages >> (stage - 1)) & 1) && a.fpad && a.view_aggregation && a.V > 1 &&
           a.w % tw == 0 && a.h % th == 0 && a.planes.D % dch == 0;
  }
  hipStream_t side_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_feat2_ = nullptr, ev_feat3_ = nullptr;
  bool side_enabled_ = true;
  size_t fork_lo_ = 0, fork_hi_ = 0, feat2_op_ = 0;  // ops [fork_lo_, fork_hi_) = fn.skip2 .. fn.out3
  // key-frame feature cache (build_fn1 .. set_batch_vfeat)
  int fcache_cap_ = 0;             // entries (0: off, the default)
  std::vector<FcEntry> fcache_;
  std::vector<Op> ops1_;           // FeatureNet for ONE view
  std::vector<FnOut> fn1_out_;
  std::string tprefix_;            // tensor-name prefix while ops1_ is built
  bool fn1_ok_ = false;            // the single-view plan exists and matches the batch plan's instances
  bool fc_fast_ = false, fc_fill_ = false;  // the staged window: answered by the cache (at most one view computed) / computed as a batch whose outputs fill the cache
  int fc_miss_ = -1, fc_slot_[kMaxSrc + 1] = {};
  int *fc_flag_ = nullptr;         // page-locked: raised by k_cache_io
  hipStream_t up_stream_ = nullptr;  // uploads of the images the cache answers (prelaunch, stage_inputs)
  hipEvent_t ev_hits_ = nullptr;
  bool prelaunched_ = false;       // the staged window's forward is already on the stream (the worker only publishes)
  bool defer_cache_io_ = false;    // ... and its image comparison follows once the uploads are in flight
  uint64_t fc_clock_ = 0, fc_hits_ = 0, fc_misses_ = 0, fc_batch_windows_ = 0, fc_collisions_ = 0;

  const MvsSwitches sw_;  // read once, here
#ifdef DR_PARITY_HOOKS
  static constexpr bool kParityHooks = true;
#else
  static constexpr bool kParityHooks = false;
#endif
  int device_;
  int *march_err_ = nullptr;
  // called after a stream synchronise: a marching convolution that gave up a wait leaves garbage behind -- report it, never return it
  void check_march() {
    if (march_err_ && *march_err_) { const int c = *march_err_; *march_err_ = 0; fail(DR_ERR_DEVICE, "k_conv_m: a ring wait gave up (code %d): the LDS producer/consumer protocol stalled", c); }
  }
  Blob blob_;
  hipStream_t stream_ = nullptr;
  DeviceArena consts_;
  const float *lut_ = nullptr;
  std::unique_ptr<DeviceArena> plan_arena_;
  std::map<std::string, DevTensor> tensors_;
  std::vector<Op> ops_;
  std::vector<void *> misc_;
  CostVolArgs cv_[3];
  RegressArgs rg_[3];
  uint8_t *d_bgr_ = nullptr, *h_in_ = nullptr;
  float *h_out_[2] = {nullptr, nullptr}, *h_out_dev_[2] = {nullptr, nullptr};  // two pinned result blocks (4 maps each), used alternately, and the addresses the device uses for them
  int out_cur_ = 0;                // the block the last result is in
  hipEvent_t ev_h2d_ = nullptr;    // completion of a window uploaded straight from the caller's page-locked images
  unsigned *d_state_ = nullptr, *d_hist_ = nullptr;
  unsigned filter_rank_ = 0;
  bool prob_fused_last_[3] = {false, false, false};  // (reporting: the last forward ran this stage's prob as k_prob2_regress)
  bool regress_done_[3] = {false, false, false};  // set by a PROB op that also ran its stage's regression, consumed by the REGRESS op behind it
  int H_ = 0, W_ = 0, V_ = 0;
  int shard_nsrc_ = 0;  // > 0: view-shard rank, cost-volume divisor = source views of the whole window
  ncclComm_t comm_ = nullptr;  // view-shard communicator (drm_comm_init); the volumes are reduced in stream order when set
  int comm_world_ = 0, comm_rank_ = 0;
  // Collective form of a sharded forward (MvsSwitches::shard_allreduce).  Default: the partial volumes are REDUCED to rank 0, which
  // alone regularises and regresses the stage, and the stage's depth map (what the next stage's hypotheses hang on: 77 / 307 /
  // 1229 KB) is BROADCAST back -- (n-1)/n x 354 MB over xGMI per depth map instead of the all-reduce's 2(n-1)/n, and no redundant
  // CostRegNet on the other ranks.  DR_SHARD_ALLREDUCE=1: round 2's form (sum all-reduce, every rank
  // regularises redundantly; no broadcast).
  bool phase_mode_ = false;

  std::thread worker_;
  HostCopier copier_;
  std::mutex mu_;
  std::condition_variable input_cv_, done_cv_;
  bool running_ = true;
  std::atomic<bool> unprocessed_{false};  // read without the lock by ready() (dr_mvsnet.cpp:54 does the same with a plain bool)
  bool has_output_ = false;
  std::string worker_error_;
};

}  // namespace dr

// ==================================================================== C ABI
using dr::guarded;
struct drm_s {
  std::unique_ptr<dr::MvsEngine> e;
};
// every C-ABI entry point goes through this: a NULL handle is an argument error, not a crash
static inline dr::MvsEngine *eng(drm_s *h) {
  if (!h || !h->e) dr::fail(DR_ERR_ARG, "NULL handle");
  return h->e.get();
}

extern "C" {

const char *dr_last_error(void) { return dr::last_error_slot().c_str(); }
const char *dr_version(void) { return "dr_mi355x 0.1 gfx950"; }

int drm_create(const char *weights_path, int device, drm_t **out) {
  return guarded([&] {
    if (!weights_path || !out) dr::fail(DR_ERR_ARG, "drm_create: null argument");
    auto *h = new drm_s();
    try { h->e.reset(new dr::MvsEngine(weights_path, device)); } catch (...) { delete h; throw; }
    *out = h;
  });
}
void drm_destroy(drm_t *h) { delete h; }
int drm_set_feature_cache(drm_t *h, int capacity) { return guarded([&] { eng(h)->set_feature_cache(capacity); }); }
int drm_feature_cache_stats(drm_t *h, uint64_t out[6]) {
  return guarded([&] { if (!out) dr::fail(DR_ERR_ARG, "drm_feature_cache_stats: null argument"); eng(h)->feature_cache_stats(out); });
}
int drm_call_async(drm_t *h, int height, int width, int view_num, int ref_index, const uint8_t *const *bgrs, const float *K9,
                   const float *const *c2ws, float depth_min, float depth_max, float discard_percentage) {
  return guarded([&] { eng(h)->call_async(height, width, view_num, ref_index, bgrs, K9, c2ws, depth_min, depth_max, discard_percentage); });
}
int drm_ready(drm_t *h) { return (h && h->e && h->e->ready()) ? 1 : 0; }
int drm_wait(drm_t *h) { return guarded([&] { eng(h)->wait(); }); }
int drm_get_result(drm_t *h, float *depth, float *confidence, float *depth_dense, float *confidence_dense) {
  return guarded([&] { eng(h)->get_result(depth, confidence, depth_dense, confidence_dense); });
}
int drm_get_result_view(drm_t *h, const float **depth, const float **confidence, const float **depth_dense, const float **confidence_dense) {
  return guarded([&] { eng(h)->get_result_view(depth, confidence, depth_dense, confidence_dense); });
}
void *drm_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void drm_host_free(void *p) { if (p) (void)hipHostFree(p); }
int drm_upload(drm_t *h, int height, int width, int view_num, int ref_index, const uint8_t *const *bgrs, const float *K9,
               const float *const *c2ws, float depth_min, float depth_max, float discard_percentage) {
  return guarded([&] { eng(h)->upload(height, width, view_num, ref_index, bgrs, K9, c2ws, depth_min, depth_max, discard_percentage); });
}
int drm_forward(drm_t *h, int iters, float *ms_total) { return guarded([&] { eng(h)->forward_n(iters, ms_total); }); }
int drm_autotune(drm_t *h, int max_candidates, float *before_ms, float *after_ms) {
  return guarded([&] { eng(h)->autotune(max_candidates, before_ms, after_ms); });
}
int drm_set_view_shard(drm_t *h, int nsrc_total) { return guarded([&] { eng(h)->set_view_shard(nsrc_total); }); }
int drm_forward_phase(drm_t *h, int phase) { return guarded([&] { eng(h)->forward_phase(phase); }); }
int drm_comm_available(void) { return guarded([&] { (void)dr::Rccl::get(); }); }
int drm_comm_unique_id(uint8_t id[128]) {
  return guarded([&] {
    if (!id) dr::fail(DR_ERR_ARG, "drm_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    dr::Rccl &r = dr::Rccl::get();
    ncclUniqueId u;
    r.check(r.GetUniqueId(&u), "ncclGetUniqueId");
    memcpy(id, &u, 128);
  });
}
int drm_comm_init(drm_t *h, int rank, int world, const uint8_t id[128]) { return guarded([&] { eng(h)->comm_init(rank, world, id); }); }
int drm_comm_destroy(drm_t *h) { return guarded([&] { eng(h)->comm_destroy(); }); }
int drm_comm_count(drm_t *h, int *nranks) {
  return guarded([&] { if (!nranks) dr::fail(DR_ERR_ARG, "drm_comm_count: null pointer"); *nranks = eng(h)->comm_count(); });
}
int drm_device_tensor(drm_t *h, const char *name, void **dptr, size_t *nfloats) {
  return guarded([&] { if (!name) dr::fail(DR_ERR_ARG, "drm_device_tensor: null name"); eng(h)->device_tensor(name, dptr, nfloats); });
}
int drm_download(drm_t *h, float *depth, float *confidence, float *depth_dense, float *confidence_dense) {
  return guarded([&] { eng(h)->download(depth, confidence, depth_dense, confidence_dense); });
}
int drm_get_stage_output(drm_t *h, int stage, float *depth, float *confidence) {
  return guarded([&] {
    if (stage < 1 || stage > 3) dr::fail(DR_ERR_ARG, "stage must be 1..3");
    size_t n; int dims[4];
    eng(h)->get_tensor(("depth" + std::to_string(stage)).c_str(), nullptr, 0, &n, dims);
    eng(h)->get_tensor(("depth" + std::to_string(stage)).c_str(), depth, n, &n, dims);
    eng(h)->get_tensor(("conf" + std::to_string(stage)).c_str(), confidence, n, &n, dims);
  });
}
int drm_get_tensor(drm_t *h, const char *name, float *out, size_t n_max, size_t *n, int dims[4]) {
  return guarded([&] { eng(h)->get_tensor(name, out, n_max, n, dims); });
}
int drm_profile(drm_t *h, char *names, size_t names_cap, float *ms, int cap, int *count) {
  return guarded([&] {
    std::string nm; std::vector<float> t;
    eng(h)->profile(nm, t);
    if (count) *count = (int)t.size();
    if (names && names_cap) { strncpy(names, nm.c_str(), names_cap - 1); names[names_cap - 1] = 0; }
    for (int i = 0; i < (int)t.size() && i < cap; ++i) ms[i] = t[i];
  });
}
int drm_work(drm_t *h, double *flops, double *bytes) { return guarded([&] { eng(h)->work(flops, bytes); }); }

int drm_debug_conv(int device, const float *in, int D, int H, int W, int Cin, const float *weight, int Cout, int kd, int kh, int kw,
                   int sd, int sh, int sw, int transposed, const float *scale, const float *bias, int relu, const float *add,
                   int add_up2, float *out, int out_dims[3]) {
  return guarded([&] {
    using namespace dr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device) fail(DR_ERR_DEVICE, "no HIP device %d", device);
    DR_HIP(hipSetDevice(device));
    DeviceArena arena;
    arena.err_flag = arena.upload(std::vector<int>(1, 0));
    ConvLayer L;
    L.Cin = Cin; L.Cout = Cout; L.kd = kd; L.kh = kh; L.kw = kw; L.sd = sd; L.sh = sh; L.sw = sw;
    L.transposed = transposed == 1; L.weight = weight; L.relu = relu != 0;
    const bool up2 = transposed == 2;  // test hook for ConvLayer::up2: both row parities, one launch each
    if (scale) L.scale.assign(scale, scale + Cout);
    if (bias) L.bias.assign(bias, bias + Cout);
    const ConvMode mode = (!transposed && sw == 1 && Cout == 8) ? CONV_XPAIR : ((!transposed && sw == 1 && Cout == 1) ? CONV_X8 : CONV_NORMAL);
    auto cz = axis_classes(kd, sd, L.transposed, D), cy = axis_classes(kh, sh, L.transposed, H), cx = axis_classes(kw, sw, L.transposed, W);
    const int oD = L.transposed ? D * sd : cz[0].npos, oH = L.transposed ? H * sh : (up2 ? 2 * H : cy[0].npos), oW = L.transposed ? W * sw : (up2 ? 2 * W : cx[0].npos);
    std::vector<float> hin(in, in + (size_t)D * H * W * Cin);
    float *d_in = arena.upload(hin);
    const size_t on = (size_t)oD * oH * oW * Cout;
    std::vector<float> zero(on, 0.f);
    float *d_out = arena.upload(zero);
    float *d_add = nullptr;
    if (add) {
      const size_t an = add_up2 ? (size_t)oD * (oH / 2) * (oW / 2) * Cout : on;
      std::vector<float> ha(add, add + an);
      d_add = arena.upload(ha);
    }
    // DR_CONV_RANK (test hook): build the rank-th candidate of the planner's ranking instead of its first choice
    const char *rk = getenv("DR_CONV_RANK");
    ConvPlanOut P;
    for (int py = 0; py < (up2 ? 2 : 1); ++py) {
      L.up2 = up2 ? 1 + py : 0;
      P = plan_conv(L, mode, d_in, D, H, W, Cin, d_out, d_add, add_up2 ? 2 : 1, arena, rk ? atoi(rk) : 0);
      for (auto &cl : P.launches) launch_conv(cl, nullptr);
    }
    DR_HIP(hipDeviceSynchronize());
    DR_HIP(hipGetLastError());
    int march_err = 0;
    DR_HIP(hipMemcpy(&march_err, arena.err_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (march_err) fail(DR_ERR_DEVICE, "k_conv_m: a ring wait gave up (code %d)", march_err);
    if (getenv("DR_CONV_PRINT")) {
      const ConvLaunch &c = P.launches.at(0);
      fprintf(stderr, "debug_conv: %s<%d,%d,%d> nup %d tile %dx%dx%d lds %zu grid %ux%u (%d candidates)%s\n", (c.async == 2 ? (c.march.rm ? "rowmarch" : (c.march.wino ? "winomarch" : "march")) : (c.async == 4 ? "wino" : (c.async ? "async" : (c.bf3 ? "bf16x3" : "sync")))), c.ci, c.ct, c.pt,
              c.nup, c.args.TZ, c.args.TY, c.args.TXT * 16, c.lds_bytes, c.grid.x, c.grid.z, P.ncand, c.args.class_loop ? ", class loop" : "");
    }
    DR_HIP(hipMemcpy(out, d_out, on * 4, hipMemcpyDeviceToHost));
    if (out_dims) { out_dims[0] = oD; out_dims[1] = oH; out_dims[2] = oW; }
  });
}

/* Kernel unit-test hook for k_tail (tail_kernels.h): conv11 (ConvTranspose3d 16 -> 8, k 3, s 2, p 1, op 1; folded BN scale / bias; ReLU; + skip) followed by prob
 * (Conv3d 8 -> 1, k 3, p 1).  x: (D/2, h/2, w/2, 16), skip: (D, h, w, 8) channels-last; w_deconv (16, 8, 3, 3, 3), w_prob (1, 8, 3, 3, 3) torch layouts;
 * qy: quad rows of the tile (0: chosen by size), zchunk: depth planes per workgroup (0: chosen), form: 1 = k_tail_m (matrix pipe), 0 = k_tail.  out: (D, h, w) logits. */
int drm_debug_tail(int device, const float *x, const float *skip, const float *w_deconv, const float *scale8, const float *bias8, const float *w_prob, int D, int h,
                   int w, int qy, int zchunk, int form, float *out) {
  return guarded([&] {
    using namespace dr;
#ifndef DR_PARITY_HOOKS
    fail(DR_ERR_UNSUPPORTED, "drm_debug_tail: k_tail / k_tail_m are built into the parity library (libdr_mi355x_hooks.so, -DDR_PARITY_HOOKS) only");
#else
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device) fail(DR_ERR_DEVICE, "no HIP device %d", device);
    if (D <= 0 || h <= 0 || w <= 0 || (D | h | w) & 1) fail(DR_ERR_ARG, "drm_debug_tail: output dims must be positive and even");
    DR_HIP(hipSetDevice(device));
    DeviceArena arena;
    TailArgs t{};
    t.x = arena.upload(std::vector<float>(x, x + (size_t)(D / 2) * (h / 2) * (w / 2) * 16));
    t.skip = arena.upload(std::vector<float>(skip, skip + (size_t)D * h * w * 8));
    t.wd = arena.upload(tail_pack_deconv(w_deconv));
    std::vector<float> sb(scale8, scale8 + 8);
    sb.insert(sb.end(), bias8, bias8 + 8);
    t.sb = arena.upload(sb);
    t.wp = arena.upload(tail_pack_prob(w_prob));
    float *d_out = arena.upload(std::vector<float>((size_t)D * h * w, -12345.f));
    t.out = d_out; t.D = D; t.h = h; t.w = w;
    const bool mf = form == 1;  // 1: k_tail_m (matrix pipe), else k_tail (vector pipe)
    t.wmf = mf ? arena.upload(tail_pack_deconv_mfma(w_deconv)) : nullptr;
    tail_pick_tile(h, w, t.QY, t.QX, mf);
    if (qy == 4 || qy == 8 || qy == 16 || (qy == 32 && !mf)) { t.QY = qy; t.QX = 256 / qy; }
    t.zchunk = zchunk > 0 ? std::min(D, zchunk) : tail_pick_zchunk(D, h, w, t.QY, t.QX);
    launch_tail(t, nullptr);
    DR_HIP(hipDeviceSynchronize());
    DR_HIP(hipGetLastError());
    DR_HIP(hipMemcpy(out, d_out, (size_t)D * h * w * 4, hipMemcpyDeviceToHost));
#endif
  });
}

}  // extern "C"
