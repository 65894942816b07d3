// fn_front.h -- k_fn_front: FeatureNet's first block in ONE launch (round 5):
//   u8 BGR -> RGB0 / 255            (k_preprocess; dr_mvsnet.cpp:184-217)
//   conv0.0  3 -> 8, 3x3, BN, ReLU  (module.py:461-466)
//   conv0.1  8 -> 8, 3x3, BN, ReLU  (module.py:467-470)
// The three launches it replaces move 34 + (34 + 69) + (69 + 69) MB for 7 x 480 x 640 and end with 34 / 69 / 69 MB of dirty lines each (the
// next launch starts behind their write-back); this one reads the 6.5 MB of u8 pixels and writes the 69 MB of `fn.conv0.1` -- the float image and
// `fn.conv0.0` never leave the CU.
//
// One workgroup = 8 waves, persistent over its tiles (XCD k owns the k-th contiguous range of tiles, as k_conv_a).  A tile is 8 rows x 64 pixels of
// output.  Per tile:
//   1. the 12 x 68 pixels of the input around it (2-pixel halo), fetched into registers ONE TILE AHEAD (under the previous tile's two K loops; a
//      pixel as the two aligned 32-bit words around its three bytes), go through the 256-entry table (in LDS) into the image tile: one float4
//      (R, G, B, 0) per pixel -- written between conv0.1's K loop and its stores, so that the wait for the prefetch is not a wait for the stores;
//   2. conv0.0 on the matrix pipe over the 10 x 66 pixels conv0.1 needs (1-pixel halo): XPAIR rows (8 channels x 2 adjacent x, 4-wide x window,
//      K = 12 taps x 4 channels = 3 chunks), the 10 x 33 pixel pairs enumerated row-major and dealt to the waves in groups of 16 -- a B operand is
//      16 ARBITRARY positions, so the region needs no padding to a multiple of 16 pairs per row (21 groups instead of 30);
//      epilogue (folded BN, ReLU, zero outside the image = conv0.1's zero padding) into the second LDS tile, 8 channels per pixel;
//   3. conv0.1 on the matrix pipe from that tile (K = 12 taps x 8 channels = 6 chunks), 16 groups = 2 per wave; epilogue to HBM.
// The products and their order are k_conv's for the same two layers in their direct XPAIR form (chunk after chunk, one accumulator per position
// group), so the result equals that path bit for bit (tests/test_mvsnet_gpu.py::test_fused_front_equals_the_three_kernel_path); the product's
// default plan runs conv0.1 in the Winograd form, from which this differs by fp32 reassociation like every direct plan does.
#pragma once
#include <algorithm>
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

constexpr int kFrontThreads = 512, kFrontWaves = 8;
constexpr int kFrontTY = 8, kFrontTXP = 64;                         // output tile: rows x pixels
constexpr int kFrontIH = kFrontTY + 4, kFrontIW = kFrontTXP + 4;    // image tile (pixels)
constexpr int kFrontAH = kFrontTY + 2, kFrontAW = kFrontTXP + 2;    // conv0.0 tile (pixels)
constexpr int kFrontP1 = kFrontAW / 2;                              // conv0.0 pixel pairs per row (33)
constexpr int kFrontNP1 = kFrontAH * kFrontP1;                      // conv0.0 positions (330)
constexpr int kFrontG1 = (kFrontNP1 + 15) / 16;                     // groups of 16 positions (21)
constexpr int kFrontPT1 = (kFrontG1 + kFrontWaves - 1) / kFrontWaves;  // per wave (3)
constexpr int kFrontG2 = kFrontTY * kFrontTXP / 32;                 // conv0.1 groups (16)
constexpr int kFrontPT2 = kFrontG2 / kFrontWaves;                   // per wave (2)
constexpr int kFrontCIS2 = 12;                                      // LDS floats per conv0.0 pixel (8 channels + 4: spreads the b128 reads over bank slots)
constexpr int kFrontNU1 = 3, kFrontNU2 = 6;                         // K chunks of 16
constexpr int kFrontNPI = kFrontIH * kFrontIW;                      // image tile pixels (816)
constexpr int kFrontPPT = (kFrontNPI + kFrontThreads - 1) / kFrontThreads;  // pixels a thread fetches (2)
static_assert(kFrontG2 % kFrontWaves == 0 && kFrontAW % 2 == 0, "tile shape");
constexpr size_t kFrontLdsBytes = (size_t)kFrontNPI * 16 + (size_t)kFrontAH * kFrontAW * kFrontCIS2 * 4 + (size_t)(kFrontNU1 + kFrontNU2) * 64 * 16 + 256 * 4;

struct FrontArgs {
  const uint8_t *bgr;     // [V][H][W][3], 4-byte aligned, followed by at least 8 readable bytes (the kernel fetches a pixel as the two aligned words around it)
  const float *lut;       // 256 floats: float(double(b) / 255.0)
  const float4 *w1, *w2;  // packed weights [chunk][lane]: lane (i = l & 15, g = l >> 4) holds K = 16 u + 4 g .. + 3 of XPAIR row i
  const float *sb1, *sb2; // 16 scales then 16 biases per layer (row r = 8 * (x of the pair) + channel)
  float *out;             // [V][H][W][8]
  int V, H, W, tilesY, tilesX, ntiles;
};

// ---- geometry shared with the host emulation (tests/cpp/front_emul.hip) ----
// conv0.0 position n (0 .. kFrontNP1): row ry of the conv0.0 tile, pixel pair q of that row; first image-tile pixel of its 3 x 4 window
DR_HD inline void front_pos1(int n, int &ry, int &q) { ry = n / kFrontP1; q = n - ry * kFrontP1; }
DR_HD inline int front_base1(int ry, int q) { return ry * kFrontIW + 2 * q; }
DR_HD inline int front_tap1(int t) { return (t >> 2) * kFrontIW + (t & 3); }                  // tap t = 4 * ky + (x offset in the 4-wide window): image-tile pixels
DR_HD inline int front_tap2(int t) { return ((t >> 2) * kFrontAW + (t & 3)) * kFrontCIS2; }   // the same in the conv0.0 tile: LDS floats
// conv0.1 group gi (0 .. 15), lane column j: output row yt, pixel pair q2 of the 32 in a row; first conv0.0-tile pixel of its window
DR_HD inline void front_pos2(int gi, int j, int &yt, int &q2) { yt = gi >> 1; q2 = (gi & 1) * 16 + j; }
DR_HD inline int front_base2(int yt, int q2) { return (yt * kFrontAW + 2 * q2) * kFrontCIS2; }
// XPAIR weight of (tap t, input channel c, row r) of a 3x3 layer with weights w[co][cin][ky][kx]: row r = 8 * shift + co computes output x = 2 q + shift
inline float front_weight(const float *w, int Cin, int t, int c, int r) {
  const int shift = r >> 3, co = r & 7, ky = t >> 2, kx = (t & 3) - shift;
  if (kx < 0 || kx > 2 || c >= Cin) return 0.f;
  return w[(((size_t)co * Cin + c) * 3 + ky) * 3 + kx];
}
// packed [chunk][lane][4]: CI input channels per tap as the kernel walks them (CI = 4: RGB0, 4 taps per chunk; CI = 8: 2 taps per chunk)
inline std::vector<float> front_pack(const float *w, int Cin, int CI) {
  const int TPC = 16 / CI, NU = 12 / TPC;
  std::vector<float> pk((size_t)NU * 64 * 4);
  for (int u = 0; u < NU; ++u) for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) {
    const int g = l >> 4, i = l & 15, k16 = 4 * g + s;
    pk[((size_t)u * 64 + l) * 4 + s] = front_weight(w, Cin, u * TPC + k16 / CI, k16 % CI, i);
  }
  return pk;
}

#ifdef __HIPCC__
typedef float front_fx4 __attribute__((ext_vector_type(4)));

// (second launch bound = waves per SIMD: two workgroups of eight waves per CU, 128 registers each)
__global__ __launch_bounds__(kFrontThreads, 4) void k_fn_front(const FrontArgs a) {
  extern __shared__ float4 lds4[];
  float4 *img = lds4;                                                     // [IH][IW] RGB0
  float *c0 = reinterpret_cast<float *>(lds4 + kFrontNPI);                // [AH][AW][CIS2]
  float4 *w1 = reinterpret_cast<float4 *>(c0 + kFrontAH * kFrontAW * kFrontCIS2);
  float4 *w2 = w1 + kFrontNU1 * 64;
  float *lut = reinterpret_cast<float *>(w2 + kFrontNU2 * 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, g = lane >> 4;

  for (int i = tid; i < kFrontNU1 * 64; i += kFrontThreads) w1[i] = a.w1[i];
  for (int i = tid; i < kFrontNU2 * 64; i += kFrontThreads) w2[i] = a.w2[i];
  for (int i = tid; i < 256; i += kFrontThreads) lut[i] = a.lut[i];
  const float4 sc1 = *reinterpret_cast<const float4 *>(a.sb1 + 4 * g), bi1 = *reinterpret_cast<const float4 *>(a.sb1 + 16 + 4 * g);
  const float4 sc2 = *reinterpret_cast<const float4 *>(a.sb2 + 4 * g), bi2 = *reinterpret_cast<const float4 *>(a.sb2 + 16 + 4 * g);
  // The affine terms are used HERE once, so that hipcc waits for them in front of the loop: their first real use sits in a conditional block of
  // the loop, and a load whose wait is not on every path keeps an s_waitcnt vmcnt(0) in front of every later use -- behind the pixel prefetch
  // and between the two stores of the epilogue.
  asm volatile("" ::"v"(sc1.x), "v"(sc1.y), "v"(sc1.z), "v"(sc1.w), "v"(bi1.x), "v"(bi1.y), "v"(bi1.z), "v"(bi1.w));
  asm volatile("" ::"v"(sc2.x), "v"(sc2.y), "v"(sc2.z), "v"(sc2.w), "v"(bi2.x), "v"(bi2.y), "v"(bi2.z), "v"(bi2.w));

  // this workgroup's tiles (k_conv_a's order: XCD k owns the k-th contiguous range, its workgroups take them round-robin)
  const int per_xcd = (a.ntiles + 7) >> 3, xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, nw = gridDim.x >> 3;
  const int t_lo = xcd * per_xcd, t_hi = min(a.ntiles, t_lo + per_xcd);
  const int my_tiles = t_lo + wi < t_hi ? (t_hi - t_lo - wi + nw - 1) / nw : 0;
  auto origin = [&](int k, int &v, int &y0, int &x0) {
    int b = t_lo + wi + k * nw;
    const int tx = b % a.tilesX;
    b /= a.tilesX;
    v = b / a.tilesY; y0 = (b % a.tilesY) * kFrontTY; x0 = tx * kFrontTXP;
  };
  // the bytes of this thread's pixels of tile k, fetched as the two ALIGNED 32-bit words that contain them (v_alignbyte extracts B, G, R when the
  // next iteration converts them).  Nothing here may wait for the loads -- they are meant to be in flight under both K loops -- and byte loads
  // do not allow that: hipcc carries an 8-bit value across the loop edge as i8 and re-extends it (v_and 0xff) directly behind its load, i.e.
  // an s_waitcnt vmcnt in front of the K loops; a 32-bit word is left alone until it is used.  A pixel outside the image reads the first
  // words of the buffer and is replaced by table entry 0 (= 0.f, the padding value) when it is converted.
  unsigned plo[kFrontPPT], phi[kFrontPPT], pin = 0;  // pin: per pixel 1 bit "inside" (bit e) and 2 bits byte offset inside plo (bits 8 + 2e ..)
  int fy[kFrontPPT], fx[kFrontPPT], frel[kFrontPPT];  // this thread's pixels: image-tile row / column, bytes from the tile's first pixel (the same for every tile)
#pragma unroll
  for (int e = 0; e < kFrontPPT; ++e) {
    const int n = e * kFrontThreads + tid;
    fy[e] = n / kFrontIW; fx[e] = n - fy[e] * kFrontIW;
    frel[e] = n < kFrontNPI ? (fy[e] * a.W + fx[e]) * 3 : -1;
  }
  auto fetch = [&](int k) {
    int v, y0, x0;
    origin(k, v, y0, x0);
    pin = 0;
    const int o0 = ((v * a.H + y0 - 2) * a.W + x0 - 2) * 3;  // (scalar; 32-bit byte offsets from the 4-byte aligned base: scalar-base loads, the host refuses 2 GB and more)
#pragma unroll
    for (int e = 0; e < kFrontPPT; ++e) {
      const bool in = frel[e] >= 0 && (unsigned)(y0 - 2 + fy[e]) < (unsigned)a.H && (unsigned)(x0 - 2 + fx[e]) < (unsigned)a.W;
      const unsigned ob = in ? (unsigned)(o0 + frel[e]) : 0u;
      const unsigned *q = reinterpret_cast<const unsigned *>(a.bgr + (size_t)(ob & ~3u));
      plo[e] = q[0]; phi[e] = q[1];
      pin |= (in ? 1u << e : 0u) | ((ob & 3u) << (8 + 2 * e));
    }
  };

  // operand addresses that do not depend on the tile
  int b1[kFrontPT1];   // image-tile pixel of this lane's conv0.0 window, per group
  int ry1[kFrontPT1], ax1[kFrontPT1];
  bool ok1[kFrontPT1];
#pragma unroll
  for (int pt = 0; pt < kFrontPT1; ++pt) {
    const int gi = wave + kFrontWaves * pt, n = gi * 16 + j;
    int ry, q;
    front_pos1(n < kFrontNP1 ? n : 0, ry, q);
    ok1[pt] = gi < kFrontG1 && n < kFrontNP1;
    b1[pt] = front_base1(ry, q);
    ry1[pt] = ry; ax1[pt] = 2 * q + (g >> 1);
  }
  int b2[kFrontPT2], yt2[kFrontPT2], ox2[kFrontPT2];
#pragma unroll
  for (int pt = 0; pt < kFrontPT2; ++pt) {
    int yt, q2;
    front_pos2(wave * kFrontPT2 + pt, j, yt, q2);
    b2[pt] = front_base2(yt, q2) + 4 * (g & 1);
    yt2[pt] = yt; ox2[pt] = 2 * q2 + (g >> 1);
  }
  int t1[kFrontNU1], t2[kFrontNU2];  // this lane's tap of each chunk (conv0.0: tap 4 u + g; conv0.1: tap 2 u + (g >> 1))
#pragma unroll
  for (int u = 0; u < kFrontNU1; ++u) t1[u] = front_tap1(4 * u + g);
#pragma unroll
  for (int u = 0; u < kFrontNU2; ++u) t2[u] = front_tap2(2 * u + (g >> 1));

  auto convert = [&]() {  // the fetched words through the table into the image tile
#pragma unroll
    for (int e = 0; e < kFrontPPT; ++e) {
      const int n = e * kFrontThreads + tid;
      const bool in = (pin >> e) & 1u;
      const unsigned w = in ? __builtin_amdgcn_alignbyte(phi[e], plo[e], (pin >> (8 + 2 * e)) & 3u) : 0u;  // bytes B, G, R, x
      if (n < kFrontNPI) img[n] = make_float4(lut[(w >> 16) & 255u], lut[(w >> 8) & 255u], lut[w & 255u], 0.f);
    }
  };
  if (my_tiles > 0) fetch(0);
  __syncthreads();  // weights and table in place
  if (my_tiles > 0) convert();
  for (int k = 0; k < my_tiles; ++k) {
    int v, y0, x0;
    origin(k, v, y0, x0);
    __syncthreads();  // image tile complete (written at the end of the previous iteration); every wave has left the previous tile's conv0.1 (the conv0.0 tile may be overwritten)
    if (k + 1 < my_tiles) fetch(k + 1);  // in flight under both K loops

    // ---- conv0.0: K = 3 chunks (4 taps x RGB0) ----
    {
      float4 av[kFrontNU1], bv[kFrontNU1][kFrontPT1];
#pragma unroll
      for (int u = 0; u < kFrontNU1; ++u) {
        av[u] = w1[u * 64 + lane];
#pragma unroll
        for (int pt = 0; pt < kFrontPT1; ++pt) bv[u][pt] = img[b1[pt] + t1[u]];
      }
      front_fx4 acc[kFrontPT1];
#pragma unroll
      for (int pt = 0; pt < kFrontPT1; ++pt) acc[pt] = front_fx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < kFrontNU1; ++u) {
#pragma unroll
        for (int pt = 0; pt < kFrontPT1; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u][pt].x, acc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < kFrontPT1; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u][pt].y, acc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < kFrontPT1; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u][pt].z, acc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < kFrontPT1; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u][pt].w, acc[pt], 0, 0, 0);
      }
#pragma unroll
      for (int pt = 0; pt < kFrontPT1; ++pt) {
        if (!ok1[pt]) continue;
        const int gy = y0 - 1 + ry1[pt], gx = x0 - 1 + ax1[pt];
        float4 o;
        o.x = fmaxf(acc[pt][0] * sc1.x + bi1.x, 0.f);
        o.y = fmaxf(acc[pt][1] * sc1.y + bi1.y, 0.f);
        o.z = fmaxf(acc[pt][2] * sc1.z + bi1.z, 0.f);
        o.w = fmaxf(acc[pt][3] * sc1.w + bi1.w, 0.f);
        if (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) o = make_float4(0.f, 0.f, 0.f, 0.f);  // conv0.1 pads with zeros, not with conv0.0 of the padding
        *reinterpret_cast<float4 *>(c0 + (ry1[pt] * kFrontAW + ax1[pt]) * kFrontCIS2 + 4 * (g & 1)) = o;
      }
    }
    __syncthreads();  // conv0.0 tile complete; every wave has left the image tile

    // ---- conv0.1: K = 6 chunks (2 taps x 8 channels) ----
    {
      float4 av[kFrontNU2], bv[kFrontNU2][kFrontPT2];
#pragma unroll
      for (int u = 0; u < kFrontNU2; ++u) {
        av[u] = w2[u * 64 + lane];
#pragma unroll
        for (int pt = 0; pt < kFrontPT2; ++pt) bv[u][pt] = *reinterpret_cast<const float4 *>(c0 + b2[pt] + t2[u]);
      }
      front_fx4 acc[kFrontPT2];
#pragma unroll
      for (int pt = 0; pt < kFrontPT2; ++pt) acc[pt] = front_fx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < kFrontNU2; ++u) {
#pragma unroll
        for (int pt = 0; pt < kFrontPT2; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u][pt].x, acc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < kFrontPT2; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u][pt].y, acc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < kFrontPT2; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u][pt].z, acc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < kFrontPT2; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u][pt].w, acc[pt], 0, 0, 0);
      }
      // The next tile's image goes to LDS HERE -- every wave left the image tile at the barrier above -- and not at the top of the next iteration:
      // behind this tile's stores the wait for the prefetched words would be a wait for the stores as well (one counter, in order).
      if (k + 1 < my_tiles) convert();
#pragma unroll
      for (int pt = 0; pt < kFrontPT2; ++pt) {
        const int gy = y0 + yt2[pt], gx = x0 + ox2[pt];
        if (gy >= a.H || gx >= a.W) continue;
        float4 o;
        o.x = fmaxf(acc[pt][0] * sc2.x + bi2.x, 0.f);
        o.y = fmaxf(acc[pt][1] * sc2.y + bi2.y, 0.f);
        o.z = fmaxf(acc[pt][2] * sc2.z + bi2.z, 0.f);
        o.w = fmaxf(acc[pt][3] * sc2.w + bi2.w, 0.f);
        *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out) + (size_t)((unsigned)(((v * a.H + gy) * a.W + gx) * 8 + 4 * (g & 1)) * 4u)) = o;
      }
    }
  }
}

inline void launch_fn_front(const FrontArgs &a, hipStream_t st) {
  const int want = std::max(1, std::min(a.ntiles, 512));
  hipLaunchKernelGGL(k_fn_front, dim3(8 * cdiv(want, 8)), dim3(kFrontThreads), kFrontLdsBytes, st, a);
}
#endif  // __HIPCC__

}  // namespace dr
