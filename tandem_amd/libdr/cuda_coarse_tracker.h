// Header-compatible shim: `class CudaCoarseTracker` (tandem/libdr/cuda_coarse_tracker/include/public/
// cuda_coarse_tracker.h:9-35) implemented on libdr_mi355x.so's C ABI (drt_*, include/dr_mi355x.h).
//
// The reference's signatures take Eigen matrices.  They are templates here so that this header does not include
// Eigen itself: any type with operator()(row, col) (4x4 / 8x8), operator()(i) (vectors) works -- Eigen::Matrix does,
// so CoarseTracker.cpp:105,144,732,777-886 compile unchanged.  calcRes returns the caller's Vec6 type via the
// template parameter of the overload below, or fills a plain array.
// Errors: the reference throws std::runtime_error (cuda_coarse_tracker.cpp:72,102-103,359); so does the shim.
#pragma once
#include <stdexcept>
#include <string>

#include "dr_mi355x.h"

class CudaCoarseTracker {
 public:
  CudaCoarseTracker(int w, int h, float setting_huberTH, float setting_coarseCutoffTH) { check(drt_create(w, h, setting_huberTH, setting_coarseCutoffTH, 0, &impl)); }
  ~CudaCoarseTracker() { drt_destroy(impl); }
  CudaCoarseTracker(const CudaCoarseTracker &) = delete;
  CudaCoarseTracker &operator=(const CudaCoarseTracker &) = delete;

  void setK(int w, int h, float fx, float fy, float cx, float cy) { check(drt_set_k(impl, w, h, fx, fy, cx, cy)); }
  void init(int n_max_in = 0) { check(drt_init(impl, n_max_in)); }
  void free() {}  // resources are released by the destructor

  template <class Vec2>
  void setReference(int n_in, float const *pc_u_in, float const *pc_v_in, float const *pc_idepth_in, float const *pc_color_in, float ref_exposure_in,
                    Vec2 const &ref_aff_g2l_in) {
    const double aff[2] = {(double) ref_aff_g2l_in(0), (double) ref_aff_g2l_in(1)};
    check(drt_set_reference(impl, n_in, pc_u_in, pc_v_in, pc_idepth_in, pc_color_in, ref_exposure_in, aff));
  }
  void setNew(float const *dInew_in) { check(drt_set_new(impl, dInew_in)); }

  // Vec6 calcRes(refToNew, new_exposure, aff_g2l, cutoffTH): pass the result type explicitly, e.g. calcRes<Vec6>(...)
  template <class Vec6, class Mat44, class Vec2>
  Vec6 calcRes(Mat44 const &refToNew, float new_exposure, Vec2 const &aff_g2l, float cutoffTH) {
    double out[6];
    calcRes(refToNew, new_exposure, aff_g2l, cutoffTH, out);
    Vec6 r;
    for (int i = 0; i < 6; i++) r(i) = out[i];
    return r;
  }
  template <class Mat44, class Vec2>
  void calcRes(Mat44 const &refToNew, float new_exposure, Vec2 const &aff_g2l, float cutoffTH, double out6[6]) {
    double T[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T[4 * r + c] = (double) refToNew(r, c);
    const double aff[2] = {(double) aff_g2l(0), (double) aff_g2l(1)};
    check(drt_calc_res(impl, T, new_exposure, aff, cutoffTH, out6, nullptr));
  }
  template <class Mat88, class Vec8, class Vec2>
  void calcG(Mat88 &H_out, Vec8 &b_out, const float new_exposure, const Vec2 &aff_g2l) {
    double H[64], b[8];
    const double aff[2] = {(double) aff_g2l(0), (double) aff_g2l(1)};
    check(drt_calc_g(impl, H, b, new_exposure, aff, nullptr));
    for (int r = 0; r < 8; r++) { for (int c = 0; c < 8; c++) H_out(r, c) = H[8 * r + c]; b_out(r) = b[r]; }
  }
  // The dense-depth branch of CoarseTracker::setCoarseTrackingRef (CoarseTracker.cpp:655-725) on the device; returns
  // the new point count.  Call after setReference(sparse points).
  int appendDenseReference(float const *depth, float const KRKi[9], float const Kt[3], int step, bool dense_only, float const *idepth0,
                           float const *dIp0, bool device_pointers = false) {
    int n = 0;
    check(drt_append_dense_reference(impl, depth, KRKi, Kt, step, dense_only ? 1 : 0, idepth0, dIp0, device_pointers ? 1 : 0, &n));
    return n;
  }
  void synchronize() { check(drt_synchronize(impl)); }
  drt_t *c_handle() { return impl; }  // extension: the C-ABI handle (introspection hooks such as drt_get_points take it)
  void startTiming() { check(drt_start_timing(impl)); }
  float endTimingMilliseconds() { float ms = -1.f; check(drt_end_timing_ms(impl, &ms)); return ms; }

 private:
  static void check(int status) { if (status != DR_OK) throw std::runtime_error(std::string(dr_last_error())); }
  drt_t *impl = nullptr;
};
