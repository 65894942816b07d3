// TEST INFRASTRUCTURE.  C API around the reference's OWN dense-depth hand-off (CoarseTracker.cpp:654-723), whose text is pulled from the
// reference checkout at build time (oracle/extract_handoff_block.py -> oracle/_ref/obj_handoff/dense_block.inc) and compiled inside the
// function below, which declares exactly the names the block refers to (members of CoarseTracker and of FrameHessian / FrameShell that it
// reads or writes).  Pins oracle/tracker_oracle.c::trk_append_dense and, through it, the HIP kernel (tests/test_ref_handoff.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

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
struct Frame { Shell *shell; Vec3f *dIp[1]; };
}  // namespace

// depth: W*H metres (<= 0: invalid); c2w_dense / c2w_last: row-major camera-to-world of the depth map's frame / of the tracker's reference frame;
// K9: level-0 intrinsics row-major; idepth0: W*H; dIp0: W*H*3 (I, dx, dy); pc_*: capacity W*H + n0 + 1, the first n0 entries are the sparse points.
// Returns pc_n[0] as the block leaves it.  NOTE the reference pre-increments: the appended points land in slots n0 + 1 .. pc_n (slot n0 is not
// written, readers of [0, pc_n) lose the last one) -- a defect the restatement and the HIP kernel do not inherit; the test compares slot for slot.
extern "C" int ref_dense_handoff(int W, int H, const float *depth, const float *c2w_dense, const double *c2w_last, const float *K9, int step, int dense_only,
                                 const float *idepth0, const float *dIp0, int n0, float *pc_u0, float *pc_v0, float *pc_idepth0, float *pc_color0, float *KRKi_out,
                                 float *Kt_out, float *Ki_out) {
  DenseDepth dd{true, c2w_dense, depth}, *dense_depth = &dd;
  Mat44 last;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) last(r, c) = c2w_last[4 * r + c];
  Shell shell{SE3(last)};
  std::vector<Vec3f> dI((size_t)W * H);
  for (size_t i = 0; i < dI.size(); i++) dI[i] = Vec3f(dIp0[3 * i], dIp0[3 * i + 1], dIp0[3 * i + 2]);
  Frame frame{&shell, {dI.data()}}, *lastRef = &frame;
  Mat33f K[1], Ki[1];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) K[0](r, c) = K9[3 * r + c];
  {  // Ki = K^-1 as CoarseTracker::makeK does it (K[level].inverse() in float, Eigen's cofactor form for 3x3): an INPUT of the block, exported through KRKi
    const Mat33f &k = K[0];
    const float det = k(0, 0) * (k(1, 1) * k(2, 2) - k(1, 2) * k(2, 1)) - k(0, 1) * (k(1, 0) * k(2, 2) - k(1, 2) * k(2, 0)) + k(0, 2) * (k(1, 0) * k(2, 1) - k(1, 1) * k(2, 0));
    const float id = 1.f / det;
    Ki[0](0, 0) = (k(1, 1) * k(2, 2) - k(1, 2) * k(2, 1)) * id; Ki[0](0, 1) = (k(0, 2) * k(2, 1) - k(0, 1) * k(2, 2)) * id; Ki[0](0, 2) = (k(0, 1) * k(1, 2) - k(0, 2) * k(1, 1)) * id;
    Ki[0](1, 0) = (k(1, 2) * k(2, 0) - k(1, 0) * k(2, 2)) * id; Ki[0](1, 1) = (k(0, 0) * k(2, 2) - k(0, 2) * k(2, 0)) * id; Ki[0](1, 2) = (k(0, 2) * k(1, 0) - k(0, 0) * k(1, 2)) * id;
    Ki[0](2, 0) = (k(1, 0) * k(2, 1) - k(1, 1) * k(2, 0)) * id; Ki[0](2, 1) = (k(0, 1) * k(2, 0) - k(0, 0) * k(2, 1)) * id; Ki[0](2, 2) = (k(0, 0) * k(1, 1) - k(0, 1) * k(1, 0)) * id;
  }
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ki_out[3 * r + c] = Ki[0](r, c);
  int w[1] = {W}, h[1] = {H};
  const int setting_tracking_step = step;
  const bool dense_tracking_with_dense_depth_only = dense_only != 0;
  float *idepth[1] = {const_cast<float *>(idepth0)};
  int pc_n[1] = {n0};
  float *pc_u[1] = {pc_u0}, *pc_v[1] = {pc_v0}, *pc_idepth[1] = {pc_idepth0}, *pc_color[1] = {pc_color0};

#include "dense_block.inc"

  return pc_n[0];
}
