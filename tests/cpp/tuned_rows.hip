// tuned_rows.hip -- host check of tandem_amd/csrc/conv_tuned.h (run by tests/test_conv_plan.py): every measured-best row must still name a
// plan the planner can build for its layer signature.  plan_conv silently falls back to its cost model when a row matches no candidate
// (a kernel instance that was removed, a tile rule that changed), which would cost performance without failing any parity test.
#include <cstdio>
#include <random>

#include "../../tandem_amd/csrc/conv_mfma.h"

namespace dr {
std::string &last_error_slot() {
  static std::string s;
  return s;
}
}  // namespace dr
using namespace dr;

int main() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  int rows = 0, stale = 0, skipped = 0;
  for (const ConvTuned &t : kConvTuned) {
    if (t.Cin == 0) continue;  // sentinel
    ++rows;
    if (t.mode >= 8) { ++skipped; continue; }  // fused-skip rows need the engine's device pointers; the GPU suite runs that layer
    ConvLayer L;
    L.Cin = t.Cin; L.Cout = t.Cout; L.kd = t.kd; L.kh = t.kh; L.kw = t.kw; L.sd = t.sd; L.sh = t.sh; L.sw = t.sw;
    L.transposed = t.transposed == 1; L.up2 = t.transposed >= 2 ? t.transposed - 1 : 0;
    std::vector<float> w((size_t)t.Cin * t.Cout * t.kd * t.kh * t.kw);
    for (auto &v : w) v = U(rng);
    L.weight = w.data();
    // plan_conv only reads the tensors' addresses: one float each is enough
    float in = 0.f, out = 0.f;
    DeviceArena arena;
    arena.host_only = true;
    ConvPlanOut P = plan_conv(L, (ConvMode)t.mode, &in, t.inD, t.inH, t.inW, t.Cin, &out, nullptr, 0, arena, 0);
    const ConvLaunch &c = P.launches.at(0);
    const int async_ = c.async == 2 ? (c.march.rm ? 3 : (c.march.wino ? 5 : 2)) : c.async;
    const bool live = c.ci == t.ci && c.ct == t.ct && c.pt == t.pt && c.args.TZ == t.tz && c.args.TY == t.ty && c.args.TXT == t.txt && async_ == t.async_;
    if (!live) {
      ++stale;
      printf("STALE row {%d,%d, %dx%dx%d s%d%d%d t%d m%d, %dx%dx%d}: wants ci=%d ct=%d pt=%d tile %dx%dx%d async=%d, planner built ci=%d ct=%d pt=%d tile %dx%dx%d async=%d\n",
             t.Cin, t.Cout, t.kd, t.kh, t.kw, t.sd, t.sh, t.sw, t.transposed, t.mode, t.inD, t.inH, t.inW, t.ci, t.ct, t.pt, t.tz, t.ty, t.txt, t.async_, c.ci, c.ct, c.pt,
             c.args.TZ, c.args.TY, c.args.TXT, async_);
    }
  }
  printf("conv_tuned.h: %d rows, %d checked, %d fused-skip rows left to the GPU suite, %d stale\n", rows, rows - skipped, skipped, stale);
  return stale ? 1 : 0;
}
