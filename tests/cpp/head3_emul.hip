// head3_emul.hip -- host emulation of k_fn_head3 (tandem_amd/csrc/fn_head3.h), run by tests/test_conv_plan.py on the CPU.
//
// The kernel's DATA FLOW on the host with the kernel's own geometry helpers and weight packing: per tile the conv0 tile and the half-resolution
// inter2 tile (zero outside the image), per wave (= output row) and group the six chunks of term A and the twelve of term B (weight set of the
// row's parity) through a scalar model of v_mfma_f32_16x16x4_f32 into ONE accumulator, the epilogue's bias and border correction, the store
// into a padded feat3.  Compared with the definition in double:
//   feat3 = conv3x3(conv0; Wa) + conv3x3(nearest_up2(inter2); Wout) + sum_t T[t] - (the T[t] of the taps outside the image)
// What this covers: both packings (XPAIR shifts; the summed kernel rows / columns of the upsampled axes per parity), every LDS index, tile
// origins, masks at ragged edges, the padded output strides.
#include <cmath>
#include <cstdio>
#include <random>

#include "../../tandem_amd/csrc/fn_head3.h"

namespace dr {
std::string &last_error_slot() {
  static std::string s;
  return s;
}
}  // namespace dr
using namespace dr;

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
  const int H2 = H / 2, W2 = W / 2, pad = 1;
  std::vector<float> c0((size_t)V * H * W * 8), i2((size_t)V * H2 * W2 * 32), wa(8 * 8 * 9), wo(8 * 32 * 9), T(9 * 8), bias16(16);
  for (auto &x : c0) x = U(rng);
  for (auto &x : i2) x = U(rng);
  for (auto &x : wa) x = 0.3f * U(rng);
  for (auto &x : wo) x = 0.2f * U(rng);
  for (auto &x : T) x = 0.1f * U(rng);
  for (int co = 0; co < 8; ++co) { float b = 0; for (int t = 0; t < 9; ++t) b += T[t * 8 + co]; bias16[co] = bias16[8 + co] = b; }
  const std::vector<float> pka = h3_pack_a(wa.data()), pkb = h3_pack_b(wo.data());
  const int out_row = (W + 2 * pad) * 8, out_plane = (H + 2 * pad) * out_row;
  std::vector<float> outp((size_t)V * out_plane, NAN);
  float *out = outp.data() + (size_t)pad * out_row + pad * 8;

  const int tilesY = (H + kH3TY - 1) / kH3TY, tilesX = (W + kH3TXP - 1) / kH3TXP;
  std::vector<float> ta((size_t)kH3AH * kH3AW * kH3CISA), tb((size_t)kH3BH * kH3BW * kH3CISB);
  for (int v = 0; v < V; ++v) for (int ty = 0; ty < tilesY; ++ty) for (int tx = 0; tx < tilesX; ++tx) {
    const int y0 = ty * kH3TY, x0 = tx * kH3TXP, m0 = y0 >> 1, n0 = x0 >> 1;
    std::fill(ta.begin(), ta.end(), NAN);
    std::fill(tb.begin(), tb.end(), NAN);
    for (int n = 0; n < kH3NA4; ++n) {
      const int pos = n >> 1, c4 = n & 1, ay = pos / kH3AW, ax = pos - ay * kH3AW, gy = y0 - 1 + ay, gx = x0 - 1 + ax;
      for (int s = 0; s < 4; ++s)
        ta[(size_t)pos * kH3CISA + 4 * c4 + s] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? c0[(((size_t)v * H + gy) * W + gx) * 8 + 4 * c4 + s] : 0.f;
    }
    for (int n = 0; n < kH3NB4; ++n) {
      const int pos = n >> 3, c4 = n & 7, by = pos / kH3BW, bx = pos - by * kH3BW, gy = m0 - 1 + by, gx = n0 - 1 + bx;
      for (int s = 0; s < 4; ++s)
        tb[(size_t)pos * kH3CISB + 4 * c4 + s] = (gy >= 0 && gy < H2 && gx >= 0 && gx < W2) ? i2[(((size_t)v * H2 + gy) * W2 + gx) * 32 + 4 * c4 + s] : 0.f;
    }
    for (int wave = 0; wave < kH3Waves; ++wave) for (int xt = 0; xt < 2; ++xt) {
      const int yt = wave, py = wave & 1;
      float acc[64][4] = {};
      for (int u = 0; u < kH3NUA; ++u) {
        float av[64][4], bv[64][4];
        for (int lane = 0; lane < 64; ++lane) {
          const int j = lane & 15, g = lane >> 4;
          const int off = h3_base_a(yt, xt * 16 + j) + 4 * (g & 1) + h3_tap_a(2 * u + (g >> 1));
          if (off < 0 || off + 3 >= (int)ta.size()) { printf("term A operand outside the conv0 tile\n"); return 1; }
          for (int s = 0; s < 4; ++s) { av[lane][s] = pka[((size_t)u * 64 + lane) * 4 + s]; bv[lane][s] = ta[off + s]; }
        }
        mfma_chunk(av, bv, acc);
      }
      for (int u = 0; u < kH3NUB; ++u) {
        float av[64][4], bv[64][4];
        for (int lane = 0; lane < 64; ++lane) {
          const int j = lane & 15, g = lane >> 4;
          const int off = h3_base_b(yt, xt * 16 + j) + 4 * g + h3_tap_b(u >> 1) + 16 * (u & 1);
          if (off < 0 || off + 3 >= (int)tb.size()) { printf("term B operand outside the inter2 tile\n"); return 1; }
          for (int s = 0; s < 4; ++s) { av[lane][s] = pkb[(((size_t)py * kH3NUB + u) * 64 + lane) * 4 + s]; bv[lane][s] = tb[off + s]; }
        }
        mfma_chunk(av, bv, acc);
      }
      for (int lane = 0; lane < 64; ++lane) {
        const int j = lane & 15, g = lane >> 4, gy = y0 + yt, gx = x0 + 2 * (xt * 16 + j) + (g >> 1);
        if (gy >= H || gx >= W) continue;
        for (int r = 0; r < 4; ++r) {
          float o = acc[lane][r] + bias16[4 * g + r];
          if (gy == 0 || gy == H - 1 || gx == 0 || gx == W - 1) {
            float corr = 0.f;
            for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
              const int yy = gy + ky - 1, xx = gx + kx - 1;
              if (yy < 0 || yy >= H || xx < 0 || xx >= W) corr += T[(ky * 3 + kx) * 8 + 4 * (g & 1) + r];
            }
            o -= corr;
          }
          if (std::isnan(o)) { printf("an unstaged element reached a non-zero weight\n"); return 1; }
          float &dst = out[(size_t)v * out_plane + (size_t)gy * out_row + (size_t)gx * 8 + 4 * (g & 1) + r];
          if (!std::isnan(dst)) { printf("output written twice\n"); return 1; }
          dst = o;
        }
      }
    }
  }

  double worst = 0;
  for (int v = 0; v < V; ++v) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int co = 0; co < 8; ++co) {
    double s = 0;
    for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xx = x + kx - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;  // inter3's zero padding: neither term, nor the bias share of this tap
      for (int c = 0; c < 8; ++c) s += (double)wa[((co * 8 + c) * 3 + ky) * 3 + kx] * c0[(((size_t)v * H + yy) * W + xx) * 8 + c];
      for (int c = 0; c < 32; ++c) s += (double)wo[((co * 32 + c) * 3 + ky) * 3 + kx] * i2[(((size_t)v * H2 + (yy >> 1)) * W2 + (xx >> 1)) * 32 + c];
      s += T[(ky * 3 + kx) * 8 + co];
    }
    const double got = out[(size_t)v * out_plane + (size_t)y * out_row + (size_t)x * 8 + co];
    if (std::isnan(got)) { printf("output (%d,%d,%d,%d) never written\n", v, y, x, co); return 1; }
    worst = std::max(worst, std::fabs(got - s) / (1.0 + std::fabs(s)));
  }
  // the zero border of the padded tensor is not the kernel's to touch
  for (int v = 0; v < V; ++v) for (int y = -1; y <= H; ++y) for (int x = -1; x <= W; ++x) {
    if (y >= 0 && y < H && x >= 0 && x < W) continue;
    for (int c = 0; c < 8; ++c) if (!std::isnan(out[(ptrdiff_t)v * out_plane + (ptrdiff_t)y * out_row + (ptrdiff_t)x * 8 + c])) { printf("border pixel (%d,%d) written\n", y, x); return 1; }
  }
  printf("head3 %d x %d x %d: max rel err %.2e %s\n", V, H, W, worst, worst < 2e-5 ? "ok" : "FAIL");
  return worst < 2e-5 ? 0 : 1;
}

int main() {
  int fails = 0;
  fails += run(2, 32, 128, 1);   // whole tiles
  fails += run(1, 40, 96, 2);    // ragged in both directions (96 = 64 + 32, 40 = 2 * 16 + 8)
  fails += run(1, 6, 34, 3);     // smaller than one tile
  fails += run(2, 64, 64, 4);
  return fails ? 1 : 0;
}
