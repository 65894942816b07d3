// front_emul.hip -- host emulation of k_fn_front (tandem_amd/csrc/fn_front.h), run by tests/test_conv_plan.py on the CPU.
//
// The kernel's DATA FLOW executed on the host with the kernel's own geometry helpers and weight packing: per tile the image tile through the
// table, conv0.0's position groups (lane j = position 16 gi + j of the row-major pair list, lane group g = tap 4 u + g) through a scalar model
// of v_mfma_f32_16x16x4_f32, its epilogue into the second tile (rows -> (x of the pair, channel), zero outside the image), conv0.1's groups
// and its epilogue into the output tensor.  Compared with a direct evaluation of the two layers (zero padding, folded BN, ReLU) in double.
// What this covers: packing (XPAIR shifts, RGB0 padding, chunk / lane order), every LDS index, tile origins and masks at ragged image edges.
#include <cmath>
#include <cstdio>
#include <random>

#include "../../tandem_amd/csrc/fn_front.h"

namespace dr {
std::string &last_error_slot() {
  static std::string s;
  return s;
}
}  // namespace dr
using namespace dr;

// D[row][col] += sum over (g, s) of A-lane(g * 16 + row)[s] * B-lane(g * 16 + col)[s]; lane (row >> 2) * 16 + col holds D rows 4 (row >> 2) .. + 3
static void mfma_chunk(const float (&av)[64][4], const float (&bv)[64][4], float (&acc)[64][4]) {
  for (int s = 0; s < 4; ++s)
    for (int col = 0; col < 16; ++col)
      for (int row = 0; row < 16; ++row) {
        float &d = acc[(row >> 2) * 16 + col][row & 3];
        for (int g = 0; g < 4; ++g) d = std::fmaf(av[g * 16 + row][s], bv[g * 16 + col][s], d);
      }
}

static int run(int V, int H, int W, unsigned seed) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<uint8_t> bgr((size_t)V * H * W * 3);
  for (auto &b : bgr) b = (uint8_t)(rng() & 255u);
  std::vector<float> lut(256);
  for (int b = 0; b < 256; ++b) lut[b] = (float)((double)(float)b / 255.0);
  std::vector<float> wa(8 * 3 * 9), wb(8 * 8 * 9), sb1(32), sb2(32);
  for (auto &x : wa) x = U(rng);
  for (auto &x : wb) x = 0.3f * U(rng);
  for (int c = 0; c < 8; ++c) {
    const float s1 = 0.5f + 0.5f * std::fabs(U(rng)), b1 = 0.3f * U(rng), s2 = 0.5f + 0.5f * std::fabs(U(rng)), b2 = 0.3f * U(rng);
    sb1[c] = sb1[8 + c] = s1; sb1[16 + c] = sb1[24 + c] = b1;
    sb2[c] = sb2[8 + c] = s2; sb2[16 + c] = sb2[24 + c] = b2;
  }
  const std::vector<float> pk1 = front_pack(wa.data(), 3, 4), pk2 = front_pack(wb.data(), 8, 8);
  std::vector<float> out((size_t)V * H * W * 8, NAN);

  const int tilesY = (H + kFrontTY - 1) / kFrontTY, tilesX = (W + kFrontTXP - 1) / kFrontTXP;
  std::vector<float> img((size_t)kFrontNPI * 4), c0((size_t)kFrontAH * kFrontAW * kFrontCIS2);
  for (int v = 0; v < V; ++v) for (int ty = 0; ty < tilesY; ++ty) for (int tx = 0; tx < tilesX; ++tx) {
    const int y0 = ty * kFrontTY, x0 = tx * kFrontTXP;
    for (int n = 0; n < kFrontNPI; ++n) {
      const int iy = n / kFrontIW, ix = n - iy * kFrontIW, gy = y0 - 2 + iy, gx = x0 - 2 + ix;
      unsigned px = 0;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const uint8_t *p = &bgr[(((size_t)v * H + gy) * W + gx) * 3];
        px = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16);
      }
      img[n * 4 + 0] = lut[(px >> 16) & 255u]; img[n * 4 + 1] = lut[(px >> 8) & 255u]; img[n * 4 + 2] = lut[px & 255u]; img[n * 4 + 3] = 0.f;
    }
    std::fill(c0.begin(), c0.end(), NAN);
    for (int wave = 0; wave < kFrontWaves; ++wave) for (int pt = 0; pt < kFrontPT1; ++pt) {
      const int gi = wave + kFrontWaves * pt;
      float acc[64][4] = {};
      for (int u = 0; u < kFrontNU1; ++u) {
        float av[64][4], bv[64][4];
        for (int lane = 0; lane < 64; ++lane) {
          const int j = lane & 15, g = lane >> 4, n = gi * 16 + j;
          int ry, q;
          front_pos1(n < kFrontNP1 ? n : 0, ry, q);
          const int idx = front_base1(ry, q) + front_tap1(4 * u + g);
          if (idx < 0 || idx >= kFrontNPI) { printf("conv0.0 operand outside the image tile\n"); return 1; }
          for (int s = 0; s < 4; ++s) { av[lane][s] = pk1[((size_t)u * 64 + lane) * 4 + s]; bv[lane][s] = img[idx * 4 + s]; }
        }
        mfma_chunk(av, bv, acc);
      }
      for (int lane = 0; lane < 64; ++lane) {
        const int j = lane & 15, g = lane >> 4, n = gi * 16 + j;
        if (gi >= kFrontG1 || n >= kFrontNP1) continue;
        int ry, q;
        front_pos1(n, ry, q);
        const int ax = 2 * q + (g >> 1), gy = y0 - 1 + ry, gx = x0 - 1 + ax;
        for (int r = 0; r < 4; ++r) {
          float o = std::max(acc[lane][r] * sb1[4 * g + r] + sb1[16 + 4 * g + r], 0.f);
          if (gy < 0 || gy >= H || gx < 0 || gx >= W) o = 0.f;
          c0[(size_t)(ry * kFrontAW + ax) * kFrontCIS2 + 4 * (g & 1) + r] = o;
        }
      }
    }
    for (int wave = 0; wave < kFrontWaves; ++wave) for (int pt = 0; pt < kFrontPT2; ++pt) {
      const int gi = wave * kFrontPT2 + pt;
      float acc[64][4] = {};
      for (int u = 0; u < kFrontNU2; ++u) {
        float av[64][4], bv[64][4];
        for (int lane = 0; lane < 64; ++lane) {
          const int j = lane & 15, g = lane >> 4;
          int yt, q2;
          front_pos2(gi, j, yt, q2);
          const int off = front_base2(yt, q2) + 4 * (g & 1) + front_tap2(2 * u + (g >> 1));
          if (off < 0 || off + 3 >= kFrontAH * kFrontAW * kFrontCIS2) { printf("conv0.1 operand outside the conv0.0 tile\n"); return 1; }
          for (int s = 0; s < 4; ++s) { av[lane][s] = pk2[((size_t)u * 64 + lane) * 4 + s]; bv[lane][s] = c0[off + s]; }
        }
        mfma_chunk(av, bv, acc);
      }
      for (int lane = 0; lane < 64; ++lane) {
        const int j = lane & 15, g = lane >> 4;
        int yt, q2;
        front_pos2(gi, j, yt, q2);
        const int gy = y0 + yt, gx = x0 + 2 * q2 + (g >> 1);
        if (gy >= H || gx >= W) continue;
        for (int r = 0; r < 4; ++r) {
          const float o = std::max(acc[lane][r] * sb2[4 * g + r] + sb2[16 + 4 * g + r], 0.f);
          if (std::isnan(o)) { printf("an unwritten conv0.0 pixel reached a non-zero weight\n"); return 1; }
          float &dst = out[(((size_t)v * H + gy) * W + gx) * 8 + 4 * (g & 1) + r];
          if (!std::isnan(dst)) { printf("output written twice\n"); return 1; }
          dst = o;
        }
      }
    }
  }

  // the two layers as defined (module.py:461-470): zero padding, folded BN, ReLU
  std::vector<double> mid((size_t)V * H * W * 8);
  auto pix = [&](int v, int y, int x, int c) -> double {
    if (y < 0 || y >= H || x < 0 || x >= W) return 0.0;
    return lut[bgr[(((size_t)v * H + y) * W + x) * 3 + (2 - c)]];
  };
  for (int v = 0; v < V; ++v) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int co = 0; co < 8; ++co) {
    double s = 0;
    for (int c = 0; c < 3; ++c) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) s += (double)wa[((co * 3 + c) * 3 + ky) * 3 + kx] * pix(v, y + ky - 1, x + kx - 1, c);
    mid[(((size_t)v * H + y) * W + x) * 8 + co] = std::max(s * sb1[co] + sb1[16 + co], 0.0);
  }
  double worst = 0;
  for (int v = 0; v < V; ++v) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int co = 0; co < 8; ++co) {
    double s = 0;
    for (int c = 0; c < 8; ++c) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xx = x + kx - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      s += (double)wb[((co * 8 + c) * 3 + ky) * 3 + kx] * mid[(((size_t)v * H + yy) * W + xx) * 8 + c];
    }
    const double ref = std::max(s * sb2[co] + sb2[16 + co], 0.0), got = out[(((size_t)v * H + y) * W + x) * 8 + co];
    if (std::isnan(got)) { printf("output (%d,%d,%d,%d) never written\n", v, y, x, co); return 1; }
    worst = std::max(worst, std::fabs(got - ref) / (1.0 + std::fabs(ref)));
  }
  printf("front %d x %d x %d: max rel err %.2e %s\n", V, H, W, worst, worst < 2e-5 ? "ok" : "FAIL");
  return worst < 2e-5 ? 0 : 1;
}

int main() {
  int fails = 0;
  fails += run(2, 16, 128, 1);   // whole tiles
  fails += run(1, 21, 96, 2);    // ragged in both directions (96 = 64 + 32, 21 = 2 * 8 + 5)
  fails += run(1, 7, 34, 3);     // smaller than one tile
  fails += run(3, 64, 64, 4);
  return fails ? 1 : 0;
}
