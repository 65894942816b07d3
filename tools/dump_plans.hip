// tools/dump_plans.hip -- host tool: the launch plan of every CostRegNet layer at the headline shapes (640x480, planes 48/32/8), as plan_conv builds it with the
// committed plan table: kernel form, channel pass width, tile, passes, K chunks, workgroups, LDS.  No device needed.
//   hipcc --offload-arch=gfx950 -O1 -std=c++17 tools/dump_plans.hip -o /tmp/dump_plans && /tmp/dump_plans
#include <cstdio>
#include "../tandem_amd/csrc/conv_mfma.h"
namespace dr { std::string &last_error_slot() { static std::string s; return s; } }
using namespace dr;

static void layer(const char *name, int D, int H, int W, int Cin, int Cout, int sd, int shw, bool tr, ConvMode mode) {
  ConvLayer L;
  L.Cin = Cin; L.Cout = Cout; L.kd = 3; L.kh = 3; L.kw = 3; L.sd = sd; L.sh = shw; L.sw = shw; L.transposed = tr;
  std::vector<float> w((size_t)Cin * Cout * 27, 0.01f);
  L.weight = w.data();
  float in = 0, out = 0;
  DeviceArena arena; arena.host_only = true;
  ConvPlanOut P = plan_conv(L, mode, &in, D, H, W, Cin, &out, nullptr, 0, arena, 0);
  const ConvLaunch &c = P.launches.at(0);
  const char *kind = c.async == 2 ? (c.march.rm ? "rowmarch" : (c.march.wino ? "winomarch" : "march")) : (c.async == 4 ? "k_conv_w" : (c.async ? "k_conv_a" : "k_conv"));
  printf("%-10s %2dx%3dx%3d %2d->%2d s%d%s  %-9s ci=%2d ct=%d pt=%d tile %dx%dx%-3d npass=%d grid=%ux%ux%u = %5u WG  lds=%3zu KB (%zu B, nuMax %d, halo %dx%dx%d)  %.2f GFLOP\n", name, D, H, W, Cin, Cout, sd, tr ? "T" : " ", kind, c.ci, c.ct,
         c.pt, c.args.TZ, c.args.TY, c.args.TXT * 16, c.args.npass, c.grid.x, c.grid.y, c.grid.z, c.grid.x * c.grid.y * c.grid.z, c.lds_bytes >> 10, c.lds_bytes, c.args.nuMax, c.args.TZI, c.args.TYI, c.args.TXI, c.flops / 1e9);
}
int main() {
  const int Ds[3] = {48, 32, 8}, hs[3] = {120, 240, 480}, ws[3] = {160, 320, 640}, Cs[3] = {32, 16, 8};
  for (int s = 0; s < 3; ++s) {
    const int D = Ds[s], h = hs[s], w = ws[s];
    char n[32];
    auto nm = [&](const char *l) { snprintf(n, sizeof n, "s%d.%s", s + 1, l); return n; };
    layer(nm("conv0"), D, h, w, Cs[s], 8, 1, 1, false, CONV_XPAIR);
    layer(nm("conv1"), D, h, w, 8, 16, 2, 2, false, CONV_NORMAL);
    layer(nm("conv2"), D / 2, h / 2, w / 2, 16, 16, 1, 1, false, CONV_NORMAL);
    layer(nm("conv3"), D / 2, h / 2, w / 2, 16, 32, 2, 2, false, CONV_NORMAL);
    layer(nm("conv4"), D / 4, h / 4, w / 4, 32, 32, 1, 1, false, CONV_NORMAL);
    layer(nm("conv5"), D / 4, h / 4, w / 4, 32, 64, 2, 2, false, CONV_NORMAL);
    layer(nm("conv6"), D / 8, h / 8, w / 8, 64, 64, 1, 1, false, CONV_NORMAL);
    layer(nm("conv7"), D / 8, h / 8, w / 8, 64, 32, 2, 2, true, CONV_NORMAL);
    layer(nm("conv9"), D / 4, h / 4, w / 4, 32, 16, 2, 2, true, CONV_NORMAL);
    layer(nm("conv11"), D / 2, h / 2, w / 2, 16, 8, 2, 2, true, CONV_NORMAL);
  }
  return 0;
}
