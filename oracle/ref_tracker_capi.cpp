// TEST INFRASTRUCTURE (oracle/): C entry points over the REFERENCE's own dense-tracker kernels, compiled for the host
// from ref:tandem/libdr/cuda_coarse_tracker/src/cuda_coarse_tracker_private.cu (calcResKernelNew :40-214, calcGKernel
// :260-372, launchers :216-237, :460-490) by oracle/Makefile.ref.  The host class (cuda_coarse_tracker.cpp) needs
// Sophus/Eigen, nvToolsExt and cnpy, none of which exist here: its few lines of input preparation stay restated in
// oracle/tracker_oracle.c, whose kernels' inputs are handed to these entry points unchanged.
#include <cstring>
#include "cuda_coarse_tracker_private.h"

extern "C" int cpu_launch_reverse_threads;

extern "C" {

// warped7: u v dx dy idepth residual weight, n floats each; outputs7: E numTermsInE numTermsInWarped numSaturated sumSquaredShiftT sumSquaredShiftRT sumSquaredShiftNum
void reftrk_calc_res(float huber, int w, int h, float fx, float fy, float cx, float cy, const float* refToNew16,
                     const float* Ki9, float aff_a, float aff_b, float maxEnergy, float cutoffTH, int n, const float* pc_u,
                     const float* pc_v, const float* pc_idepth, const float* pc_color, const float* dInew, float* const* warped7,
                     float* outputs7) {
  memset(outputs7, 0, 7 * sizeof(float));
  cpu_launch_reverse_threads = 1;   // thread 0 publishes the block sums: it has to run last (ref_stub/cub/block/block_reduce.cuh)
  float2 aff; aff.x = aff_a; aff.y = aff_b;
  callCalcResKernel(128, nullptr, huber, w, h, fx, fy, cx, cy, refToNew16, Ki9, aff, maxEnergy, cutoffTH, n, pc_u, pc_v,
                    pc_idepth, pc_color, dInew, warped7[0], warped7[1], warped7[2], warped7[3], warped7[4], warped7[5],
                    warped7[6], outputs7);
  cpu_launch_reverse_threads = 0;
}

void reftrk_calc_g_float(float fx, float fy, float aff_a, float aff_b, float lastRef_aff_g2l_b, int n, int loops,
                         const float* pc_color, float* const* warped7, float* outputs45) {
  memset(outputs45, 0, 45 * sizeof(float));
  cpu_launch_reverse_threads = 1;
  float2 aff; aff.x = aff_a; aff.y = aff_b;
  callCalcGKernel<float>(128, nullptr, fx, fy, aff, lastRef_aff_g2l_b, n, loops, pc_color, warped7[0], warped7[1], warped7[2],
                         warped7[3], warped7[4], warped7[5], warped7[6], outputs45);
  cpu_launch_reverse_threads = 0;
}

void reftrk_calc_g_double(float fx, float fy, float aff_a, float aff_b, float lastRef_aff_g2l_b, int n, int loops,
                          const float* pc_color, float* const* warped7, double* outputs45) {
  memset(outputs45, 0, 45 * sizeof(double));
  cpu_launch_reverse_threads = 1;
  float2 aff; aff.x = aff_a; aff.y = aff_b;
  callCalcGKernel<double>(128, nullptr, fx, fy, aff, lastRef_aff_g2l_b, n, loops, pc_color, warped7[0], warped7[1], warped7[2],
                          warped7[3], warped7[4], warped7[5], warped7[6], outputs45);
  cpu_launch_reverse_threads = 0;
}

}  // extern "C"
