// Test driver (tests/test_handoff_edit.py): compiles tandem_amd/libdr/patches/CoarseTracker_dense_handoff.inc -- the edit of
// CoarseTracker::setCoarseTrackingRef that INTEGRATION.md describes -- inside the same frame of member names the reference's own block is
// compiled in (oracle/ref_handoff_capi.cpp), against the header-compatible shim tandem_amd/libdr/cuda_coarse_tracker.h, and returns the point
// list the tracker ends up with.  The Eigen / Sophus types come from oracle/ref_stub_eigen/handoff_types.h (absent from the image).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "cuda_coarse_tracker.h"
#include "handoff_types.h"

#define HANDOFF_EXPORT(KRKi, Kt)                                                     \
  do {                                                                               \
    for (int r_ = 0; r_ < 3; r_++) {                                                 \
      for (int c_ = 0; c_ < 3; c_++) KRKi_out[3 * r_ + c_] = KRKi(r_, c_);           \
      Kt_out[r_] = Kt[r_];                                                           \
    }                                                                                \
  } while (0)

namespace {
struct DenseDepth { bool is_valid; const float *cam_to_world; const float *depth; };
struct Shell { SE3 camToWorld; };
struct Frame { Shell *shell; Vec3f *dIp[1]; float ab_exposure; };
struct Aff { double a, b; struct V { double v[2]; double operator()(int i) const { return v[i]; } }; V vec() const { return V{{a, b}}; } };
}  // namespace

extern "C" int edit_dense_handoff(int W, int H, const float *depth, const float *c2w_dense, const double *c2w_last, const float *K9, const float *Ki9, int step,
                                  int dense_only, const float *idepth0, const float *dIp0, int n0, float *pc_u0, float *pc_v0, float *pc_idepth0, float *pc_color0,
                                  int cap, float *KRKi_out, float *Kt_out) {
  try {
    CudaCoarseTracker tracker(W, H, 9.f, 20.f), *cudaCoarseTracker = &tracker;
    tracker.setK(W, H, K9[0], K9[4], K9[2], K9[5]);
    tracker.init();
    DenseDepth dd{true, c2w_dense, depth}, *dense_depth = depth ? &dd : nullptr;  // depth == nullptr: a reference frame without a rendered depth map
    const bool dense_depth_on_device = false;
    Mat44 last;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) last(r, c) = c2w_last[4 * r + c];
    Shell shell{SE3(last)};
    std::vector<Vec3f> dI((size_t)W * H);
    for (size_t i = 0; i < dI.size(); i++) dI[i] = Vec3f(dIp0[3 * i], dIp0[3 * i + 1], dIp0[3 * i + 2]);
    Frame frame{&shell, {dI.data()}, 1.f}, *lastRef = &frame;
    Aff lastRef_aff_g2l{0.0, 0.0};
    Mat33f K[1], Ki[1];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { K[0](r, c) = K9[3 * r + c]; Ki[0](r, c) = Ki9[3 * r + c]; }
    const int setting_tracking_step = step;
    const bool dense_tracking_with_dense_depth_only = dense_only != 0;
    float *idepth[1] = {const_cast<float *>(idepth0)};
    int pc_n[1] = {n0};
    float *pc_u[1] = {pc_u0}, *pc_v[1] = {pc_v0}, *pc_idepth[1] = {pc_idepth0}, *pc_color[1] = {pc_color0};

#include "CoarseTracker_dense_handoff.inc"
    // (the reference's function ends here once its :732 is deleted, as the .inc's header says: nothing re-uploads the host arrays)

    int n = 0;
    if (drt_get_points(tracker.c_handle(), pc_u0, pc_v0, pc_idepth0, pc_color0, cap, &n) != DR_OK || n != pc_n[0]) return -2;
    return n;
  } catch (const std::exception &e) {
    fprintf(stderr, "edit_dense_handoff: %s\n", e.what());
    return -1;
  }
}
