// A caller written against the reference's CudaCoarseTracker API (cuda_coarse_tracker.h:9-35) with minimal
// Eigen-like matrix types, compiled with plain g++ against the shim header: one reference/new pair on a linear ramp
// image, whose residual is known in closed form.
#include <cmath>
#include <cstdio>
#include <vector>

#include "cuda_coarse_tracker.h"

template <int R, int C>
struct Mat {
  double d[R * C] = {};
  double &operator()(int r, int c) { return d[r * C + c]; }
  double operator()(int r, int c) const { return d[r * C + c]; }
  double &operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
};

int main() {
  const int w = 64, h = 48;
  CudaCoarseTracker t(w, h, 9.0f, 20.0f);
  t.setK(w, h, 50.f, 40.f, 31.5f, 23.5f);
  t.init();
  std::vector<float> dI(3 * w * h);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { float *p = &dI[3 * (x + y * w)]; p[0] = 0.5f * x - 0.25f * y + 100.f; p[1] = 0.5f; p[2] = -0.25f; }
  float u = 20.f, v = 30.f, id = 0.5f, col = 117.f;
  Mat<2, 1> aff;
  t.setReference(1, &u, &v, &id, &col, 1.f, aff);
  t.setNew(dI.data());
  Mat<4, 4> T;
  for (int i = 0; i < 4; i++) T(i, i) = 1;
  T(0, 3) = 0.2;
  Mat<6, 1> res = t.calcRes<Mat<6, 1>>(T, 1.f, aff, 20.f);
  const double Ku = 50.0 * ((20 - 31.5) / 50.0 + 0.1) + 31.5, r = 0.5 * Ku - 0.25 * 30 + 100 - 117, hw = 9.0 / std::fabs(r);
  const double E = hw * r * r * (2 - hw);
  Mat<8, 8> H;
  Mat<8, 1> b;
  t.calcG(H, b, 1.f, aff);
  printf("tracker: E=%g (expect %g) terms=%g H00=%g b7=%g\n", res(0), E, res(1), H(0, 0), b(7));
  bool ok = std::fabs(res(0) - E) < 1e-3 * E && res(1) == 1 && H(0, 0) > 0 && std::fabs(b(7) - (-1000.0 * hw * r)) < 1e-2 * std::fabs(1000.0 * hw * r);
  try { t.init(); ok = false; } catch (const std::runtime_error &) {}  // "Cannot call init more than once" (cuda_coarse_tracker.cpp:103)
  return ok ? 0 : 1;
}
