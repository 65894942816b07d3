// fn_head3.h -- k_fn_head3: FeatureNet's stage-3 head in ONE launch (round 5).
//   feat3 = out.stage3( up2(inter2) + skip.stage3(conv0) )                                   (module.py:480-485, 524-529)
// is linear (no BatchNorm, no ReLU, no bias in out.stage3), and since round 3 it is evaluated in the folded form
//   conv3x3(conv0; Wout . Wskip)  +  conv3x3(up2(inter2); Wout)  +  (Wout . bskip, corrected at the image border)
// as an 8 -> 8 XPAIR layer at full resolution (fn.out3a), two half-resolution phase layers that add their rows in place (fn.out3b/c:
// ConvLayer::up2) and a border kernel (fn.out3d): four launches that move 413 MB for 7 x 480 x 640 -- `feat3` is written once and
// read-modified-written twice, `inter2` is read twice -- and each ends with tens of MB of dirty lines the next one starts behind.
// Here the three terms meet in ONE accumulator: 69 MB of conv0 + 69 MB of inter2 in, 69 MB of feat3 out.
//
// One workgroup = 16 waves (one per CU: 134 KB of LDS), persistent over its tiles (XCD k owns the k-th contiguous range).  A tile is 16 rows x 64
// pixels of feat3; wave w computes row w as two groups of 16 pixel PAIRS (XPAIR rows: 8 channels x 2 adjacent x).  Per tile:
//   1. the conv0 tile (18 x 66 pixels x 8 channels, 1-pixel halo) and the inter2 tile (10 x 34 half-resolution pixels x 32 channels, 1-pixel
//      halo), fetched into registers ONE TILE AHEAD (under the previous tile's K loops), are written to LDS;
//   2. term A: K = 12 taps (3 rows x the 4-wide x window of a pair) x 8 channels = 6 chunks from the conv0 tile;
//   3. term B into the SAME accumulators: pair q of row y = 2 m + py IS half-resolution pixel (m, q), its two pixels are the two x parities,
//      so the rows are (x parity, channel) exactly as in term A.  Row parity py reads half-resolution rows {m - 1, m} (py = 0) or {m, m + 1}
//      (py = 1) with the kernel rows that fall on the same input row summed (up2_kernel_set), the x axis reads {q - 1, q, q + 1} with the
//      kernel columns summed per x parity (zero where a parity does not use an offset): K = 6 taps x 32 channels = 12 chunks, one weight set
//      per row parity (wave-uniform: py = w & 1);
//   4. epilogue: + the interior bias, - the taps of it that fall outside the image (border pixels only), one store into the padded feat3.
// feat3 differs from the four-launch form by fp32 reassociation (the partial sums of the three terms are added in another order, and
// fn.out3a runs in the Winograd form there): tests/test_mvsnet_gpu.py::test_fused_head3_equals_the_four_launch_form holds it to 5e-6 of
// the tensor's range; tests/cpp/head3_emul.hip runs the kernel's data flow on the host against the definition in double.
#pragma once
#include <algorithm>
#include <atomic>
#include <vector>

#include "dr_common.h"
#ifndef DR_HD
#if defined(__HIPCC__)
#define DR_HD __host__ __device__
#else
#define DR_HD
#endif
#endif

namespace dr {

constexpr int kH3Threads = 1024, kH3Waves = 16;
constexpr int kH3TY = 16, kH3TXP = 64;                              // output tile: rows x pixels (one row per wave)
constexpr int kH3AH = kH3TY + 2, kH3AW = kH3TXP + 2, kH3CISA = 12;  // conv0 tile: pixels, LDS floats per pixel (8 channels + 4)
constexpr int kH3BH = kH3TY / 2 + 2, kH3BW = kH3TXP / 2 + 2, kH3CISB = 36;  // inter2 tile: half-resolution pixels, LDS floats per pixel (32 + 4)
constexpr int kH3NUA = 6, kH3NUB = 12;                              // K chunks of 16: term A (2 taps x 8 channels each), term B (1 tap x 16 channels each)
constexpr int kH3NA4 = kH3AH * kH3AW * 2, kH3NB4 = kH3BH * kH3BW * 8;  // float4s of the two tiles in memory
constexpr int kH3PA = (kH3NA4 + kH3Threads - 1) / kH3Threads, kH3PB = (kH3NB4 + kH3Threads - 1) / kH3Threads;  // per thread (3, 3)
constexpr size_t kH3LdsBytes = (size_t)kH3AH * kH3AW * kH3CISA * 4 + (size_t)kH3BH * kH3BW * kH3CISB * 4 + (size_t)(kH3NUA + 2 * kH3NUB) * 64 * 16;

struct Head3Args {
  const float *c0;         // conv0 = fn.conv0.1: [V][H][W][8]
  const float *i2;         // inter2: [V][H/2][W/2][32]
  const float4 *wa;        // term A, packed [6][64]
  const float4 *wb;        // term B, packed [row parity][12][64]
  const float *bias16;     // interior bias per XPAIR row (16)
  const float *T;          // [9][8]: the bias share of every tap (border correction)
  float *out;              // first LOGICAL pixel of feat3 (padded tensor: strides below)
  int V, H, W;             // full resolution (H, W even); every tensor below 2 GB (32-bit byte offsets)
  int out_row, out_plane;  // floats between two rows / two views of `out`
  int tilesY, tilesX, ntiles;
};

// ---- geometry and packing shared with the host emulation (tests/cpp/head3_emul.hip) ----
// term A, lane (j, g) of group xt of row yt: first float of its window in the conv0 tile, float offset of tap t = 4 ky + (x offset in the 4-wide window)
DR_HD inline int h3_base_a(int yt, int q) { return (yt * kH3AW + 2 * q) * kH3CISA; }
DR_HD inline int h3_tap_a(int t) { return ((t >> 2) * kH3AW + (t & 3)) * kH3CISA; }
// term B: pair q of output row yt = half-resolution pixel (yt >> 1, q); its first input row in the inter2 tile is (yt >> 1) + (yt & 1)
DR_HD inline int h3_base_b(int yt, int q) { return (((yt >> 1) + (yt & 1)) * kH3BW + q) * kH3CISB; }
DR_HD inline int h3_tap_b(int t) { return ((t / 3) * kH3BW + (t % 3)) * kH3CISB; }  // tap t = 3 ty + tx (ty = 0, 1: the row parity's two input rows)
// kernel indices that (parity, input offset 0..2 relative to m - 1) of the upsampled axis sums (conv_mfma.h up2_kernel_set)
inline int h3_kset(int parity, int off, int (&k)[2]) {
  if (parity == 0) { if (off == 0) { k[0] = 0; return 1; } if (off == 1) { k[0] = 1; k[1] = 2; return 2; } return 0; }
  if (off == 1) { k[0] = 0; k[1] = 1; return 2; }
  if (off == 2) { k[0] = 2; return 1; }
  return 0;
}
// term A weights [co][c8][ky][kx] (the composed Wout . Wskip) -> [chunk u][lane][4]: lane (i = row, g), K index 4 g + s of chunk u = tap 2 u + (k16 >> 3), channel k16 & 7
inline std::vector<float> h3_pack_a(const float *wa) {
  std::vector<float> pk((size_t)kH3NUA * 64 * 4);
  for (int u = 0; u < kH3NUA; ++u) for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) {
    const int g = l >> 4, i = l & 15, k16 = 4 * g + s, t = 2 * u + (k16 >> 3), c = k16 & 7;
    const int shift = i >> 3, co = i & 7, ky = t >> 2, kx = (t & 3) - shift;
    pk[((size_t)u * 64 + l) * 4 + s] = (kx < 0 || kx > 2) ? 0.f : wa[(((size_t)co * 8 + c) * 3 + ky) * 3 + kx];
  }
  return pk;
}
// term B weights wo[co][c][ky][kx] (out.stage3) -> [row parity][chunk u][lane][4]: chunk u = tap (u >> 1) = 3 ty + tx, channels 16 (u & 1) + 4 g + s
inline std::vector<float> h3_pack_b(const float *wo) {
  std::vector<float> pk((size_t)2 * kH3NUB * 64 * 4);
  for (int py = 0; py < 2; ++py) for (int u = 0; u < kH3NUB; ++u) for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) {
    const int g = l >> 4, i = l & 15, t = u >> 1, c = 16 * (u & 1) + 4 * g + s, ty = t / 3, tx = t % 3, px = i >> 3, co = i & 7;
    int ky[2], kx[2];
    const int ny = h3_kset(py, py + ty, ky), nx = h3_kset(px, tx, kx);
    double acc = 0;
    for (int a = 0; a < ny; ++a) for (int b = 0; b < nx; ++b) acc += (double)wo[(((size_t)co * 32 + c) * 3 + ky[a]) * 3 + kx[b]];
    pk[(((size_t)py * kH3NUB + u) * 64 + l) * 4 + s] = (float)acc;
  }
  return pk;
}

#ifdef __HIPCC__
typedef float h3_fx4 __attribute__((ext_vector_type(4)));

// (second launch bound = waves per SIMD: sixteen waves per CU, 128 registers each)
__global__ __launch_bounds__(kH3Threads, 4) void k_fn_head3(const Head3Args a) {
  extern __shared__ float4 lds4[];
  float *ta = reinterpret_cast<float *>(lds4);                       // conv0 tile [AH][AW][CISA]
  float *tb = ta + kH3AH * kH3AW * kH3CISA;                          // inter2 tile [BH][BW][CISB]
  float4 *wa = reinterpret_cast<float4 *>(tb + kH3BH * kH3BW * kH3CISB);
  float4 *wb = wa + kH3NUA * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, g = lane >> 4;

  for (int i = tid; i < kH3NUA * 64; i += kH3Threads) wa[i] = a.wa[i];
  for (int i = tid; i < 2 * kH3NUB * 64; i += kH3Threads) wb[i] = a.wb[i];
  const float4 bias = *reinterpret_cast<const float4 *>(a.bias16 + 4 * g);
  asm volatile("" ::"v"(bias.x), "v"(bias.y), "v"(bias.z), "v"(bias.w));  // (waited for here, not inside the loop: see fn_front.h)

  const int per_xcd = (a.ntiles + 7) >> 3, xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, nw = gridDim.x >> 3;
  const int t_lo = xcd * per_xcd, t_hi = min(a.ntiles, t_lo + per_xcd);
  const int my_tiles = t_lo + wi < t_hi ? (t_hi - t_lo - wi + nw - 1) / nw : 0;
  auto origin = [&](int k, int &v, int &y0, int &x0) {
    int b = t_lo + wi + k * nw;
    const int tx = b % a.tilesX;
    b /= a.tilesX;
    v = b / a.tilesY; y0 = (b % a.tilesY) * kH3TY; x0 = tx * kH3TXP;
  };
  const int H2 = a.H >> 1, W2 = a.W >> 1;

  // this thread's float4s of the two tiles: tile-relative coordinates, LDS destinations and the offset from the tile's first element are the same
  // for every tile (32-bit byte offsets from the tensor's base: the loads take the scalar-base form, no 64-bit address arithmetic per element;
  // the host refuses tensors of 2 GB and more)
  int ay[kH3PA], ax[kH3PA], adst[kH3PA], arel[kH3PA];  // conv0: tile row / column, LDS float index (-1: none), floats from the tile origin
  int by[kH3PB], bx[kH3PB], bdst[kH3PB], brel[kH3PB];
#pragma unroll
  for (int e = 0; e < kH3PA; ++e) {
    const int n = e * kH3Threads + tid, pos = n >> 1, c4 = n & 1;
    ay[e] = pos / kH3AW; ax[e] = pos - ay[e] * kH3AW;
    adst[e] = n < kH3NA4 ? pos * kH3CISA + 4 * c4 : -1; arel[e] = (ay[e] * a.W + ax[e]) * 8 + 4 * c4;
  }
#pragma unroll
  for (int e = 0; e < kH3PB; ++e) {
    const int n = e * kH3Threads + tid, pos = n >> 3, c4 = n & 7;
    by[e] = pos / kH3BW; bx[e] = pos - by[e] * kH3BW;
    bdst[e] = n < kH3NB4 ? pos * kH3CISB + 4 * c4 : -1; brel[e] = (by[e] * W2 + bx[e]) * 32 + 4 * c4;
  }
  float4 pa[kH3PA], pb[kH3PB];
  auto fetch = [&](int k) {  // nothing is waited for here
    int v, y0, x0;
    origin(k, v, y0, x0);
    const int oa = ((v * a.H + y0 - 1) * a.W + x0 - 1) * 8;  // (scalar; negative for the first tile: its out-of-image elements are not loaded)
#pragma unroll
    for (int e = 0; e < kH3PA; ++e) {
      pa[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (adst[e] >= 0 && (unsigned)(y0 - 1 + ay[e]) < (unsigned)a.H && (unsigned)(x0 - 1 + ax[e]) < (unsigned)a.W)
        pa[e] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.c0) + (size_t)((unsigned)(oa + arel[e]) * 4u));
    }
    const int m0 = y0 >> 1, n0 = x0 >> 1, ob = ((v * H2 + m0 - 1) * W2 + n0 - 1) * 32;
#pragma unroll
    for (int e = 0; e < kH3PB; ++e) {
      pb[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bdst[e] >= 0 && (unsigned)(m0 - 1 + by[e]) < (unsigned)H2 && (unsigned)(n0 - 1 + bx[e]) < (unsigned)W2)
        pb[e] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.i2) + (size_t)((unsigned)(ob + brel[e]) * 4u));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int e = 0; e < kH3PA; ++e) if (adst[e] >= 0) *reinterpret_cast<float4 *>(ta + adst[e]) = pa[e];
#pragma unroll
    for (int e = 0; e < kH3PB; ++e) if (bdst[e] >= 0) *reinterpret_cast<float4 *>(tb + bdst[e]) = pb[e];
  };

  // operand addresses: wave = output row of the tile, two groups of 16 pairs
  const int yt = wave, py = wave & 1;
  int ba[2], bb[2];
#pragma unroll
  for (int xt = 0; xt < 2; ++xt) {
    ba[xt] = h3_base_a(yt, xt * 16 + j) + 4 * (g & 1);
    bb[xt] = h3_base_b(yt, xt * 16 + j) + 4 * g;
  }
  int tpa[kH3NUA];  // term A: this lane's tap of chunk u is 2 u + (g >> 1)
#pragma unroll
  for (int u = 0; u < kH3NUA; ++u) tpa[u] = h3_tap_a(2 * u + (g >> 1));
  const float4 *wbp = wb + py * kH3NUB * 64 + lane;

  if (my_tiles > 0) fetch(0);
  __syncthreads();  // weights in place
  if (my_tiles > 0) stage();
  for (int k = 0; k < my_tiles; ++k) {
    int v, y0, x0;
    origin(k, v, y0, x0);
    __syncthreads();  // both tiles complete (written at the end of the previous iteration)
    fetch(min(k + 1, my_tiles - 1));  // in flight under the K loops.  (Unconditional -- the last iteration fetches its own tile again: with the fetch and
                                      // the LDS writes below under `if (k + 1 < my_tiles)` hipcc cannot pair them and waits, at the top of every iteration,
                                      // for everything outstanding -- this tile's stores -- before it lets the fetch overwrite its registers.)

    h3_fx4 acc[2] = {h3_fx4{0.f, 0.f, 0.f, 0.f}, h3_fx4{0.f, 0.f, 0.f, 0.f}};
    // ---- term A: 6 chunks from the conv0 tile, three at a time ----
#pragma unroll
    for (int u0 = 0; u0 < kH3NUA; u0 += 3) {
      float4 av[3], bv[3][2];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        av[d] = wa[(u0 + d) * 64 + lane];
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) bv[d][xt] = *reinterpret_cast<const float4 *>(ta + ba[xt] + tpa[u0 + d]);
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d].x, bv[d][xt].x, acc[xt], 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d].y, bv[d][xt].y, acc[xt], 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d].z, bv[d][xt].z, acc[xt], 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d].w, bv[d][xt].w, acc[xt], 0, 0, 0);
      }
    }
    // ---- term B: 12 chunks from the inter2 tile (tap u >> 1, channel half u & 1), three at a time ----
#pragma unroll
    for (int u0 = 0; u0 < kH3NUB; u0 += 3) {
      float4 av[3], bv[3][2];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int u = u0 + d;
        av[d] = wbp[u * 64];
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) bv[d][xt] = *reinterpret_cast<const float4 *>(tb + bb[xt] + h3_tap_b(u >> 1) + 16 * (u & 1));
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d].x, bv[d][xt].x, acc[xt], 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d].y, bv[d][xt].y, acc[xt], 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d].z, bv[d][xt].z, acc[xt], 0, 0, 0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) acc[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[d].w, bv[d][xt].w, acc[xt], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave has left both tiles
    stage();  // the next tiles go to LDS BEFORE this tile's stores: behind them, the wait for the prefetched words would wait for the stores too

    // ---- epilogue: interior bias, border correction, one store per pixel half ----
    const int gy = y0 + yt;
    const bool edge_tile = y0 == 0 || y0 + kH3TY >= a.H || x0 == 0 || x0 + kH3TXP >= a.W;  // (wave-uniform)
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
      const int gx = x0 + 2 * (xt * 16 + j) + (g >> 1);
      if (gy >= a.H || gx >= a.W) continue;
      float4 o = make_float4(acc[xt][0] + bias.x, acc[xt][1] + bias.y, acc[xt][2] + bias.z, acc[xt][3] + bias.w);
      if (edge_tile && (gy == 0 || gy == a.H - 1 || gx == 0 || gx == a.W - 1)) {  // the taps that see inter3's ZERO padding, not the bias (k_out3_border's sum, in its order)
        float4 corr = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int yy = gy + ky - 1, xx = gx + kx - 1;
            if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) {
              const float4 t = *reinterpret_cast<const float4 *>(a.T + (ky * 3 + kx) * 8 + 4 * (g & 1));
              corr.x += t.x; corr.y += t.y; corr.z += t.z; corr.w += t.w;
            }
          }
        o.x -= corr.x; o.y -= corr.y; o.z -= corr.z; o.w -= corr.w;
      }
      *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out) + (size_t)((unsigned)(v * a.out_plane + gy * a.out_row + gx * 8 + 4 * (g & 1)) * 4u)) = o;
    }
  }
}

inline void launch_fn_head3(const Head3Args &a, hipStream_t st) {
  static std::atomic<unsigned long long> done{0};  // bit d: the LDS opt-in is set on device d
  int dev = 0;
  DR_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    DR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fn_head3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kH3LdsBytes));
    done.fetch_or(bit, std::memory_order_release);
  }
  const int want = std::max(1, std::min(a.ntiles, 256));
  hipLaunchKernelGGL(k_fn_head3, dim3(8 * cdiv(want, 8)), dim3(kH3Threads), kH3LdsBytes, st, a);
}
#endif  // __HIPCC__

}  // namespace dr
