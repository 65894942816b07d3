// mvs_kernels.h -- the non-convolution kernels of the CVA-MVSNet depth pipeline (gfx950):
//   k_preprocess : u8 BGR HWC -> f32 RGB0 channels-last          (dr_mvsnet.cpp:184-217)
//   k_costvol    : depth hypotheses + homography warp + bilinear fetch + view-aggregation gate,
//                  accumulated over all source views in registers  (module.py:764-908, :1061-1110,
//                  cva_mvsnet.py:133-152, :73-83)
//   k_regress    : softmax over D, depth expectation, 4-neighbour confidence (module.py:1116-1133)
//   k_edge / k_hist / k_scan / k_apply : 5x5 order-statistic edge filter with an exact
//                  radix-select quantile                            (module.py:1320-1361)
//   k_prob       : CostRegNet's Cout = 1 head on the vector pipe, one output column per lane marching along z  (module.py:575)
//   k_skip_up    : FeatureNet skip (1x1 conv + bias + nearest x2 of the coarser level) as a streaming kernel; the same
//                  arithmetic runs inside k_conv's fused staging (conv_mfma.h) for stage 3
// Coalesced float4 channels-last accesses, no re-materialised intermediates (the reference writes and re-reads a
// (C,D,h,w) volume ~7 times per source view); k_costvol walks the image in XCD-aware bands so that its gathers hit L2.
#pragma once
#include <climits>
#include <utility>

#include "dr_common.h"

namespace dr {

constexpr int kMaxSrc = 7;  // view_num <= 8

// ------------------------------------------------------------------ pre-processing
// lut[b] = float(double(float(b)) / 255.0)  (the reference divides in double, dr_mvsnet.cpp:212-214)
__global__ void k_preprocess(const uint8_t *__restrict__ bgr, float4 *__restrict__ img, const float *__restrict__ lut,
                             size_t npix) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const uint8_t *p = bgr + 3 * i;
  img[i] = make_float4(lut[p[2]], lut[p[1]], lut[p[0]], 0.f);
}

// Key-frame feature cache (dr_mvsnet.hip): a cache HIT is decided on the host from a sampled hash of the image; this kernel makes it exact -- the image just
// uploaded is compared, 16 bytes per lane, with the copy the cache entry keeps; any difference raises *flag (page-locked host memory, read after the
// window's stream synchronise: the engine then drops the cache and runs the window again without it).  The same launch files the window's NEW image
// (blockIdx.y = its view) in the entry its features go to.
struct CacheIoArgs {
  const uint4 *img[kMaxSrc + 1];  // the window's images on the device, model order
  uint4 *entry[kMaxSrc + 1];      // the cache entry's copy of each
  int miss;                       // view that is copied instead of compared (-1: none)
  int *flag;
  size_t n16;
};
__global__ __launch_bounds__(256) void k_cache_io(const CacheIoArgs a) {
  const int v = blockIdx.y;
  const uint4 *__restrict__ x = a.img[v];
  uint4 *__restrict__ y = a.entry[v];
  const size_t nt = (size_t)gridDim.x * blockDim.x;
  if (v == a.miss) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n16; i += nt) y[i] = x[i];
    return;
  }
  bool diff = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n16; i += nt) {
    const uint4 p = x[i], q = y[i];
    diff |= (p.x != q.x) | (p.y != q.y) | (p.z != q.z) | (p.w != q.w);
  }
  if (__ballot(diff) != 0 && (threadIdx.x & 63) == 0) *a.flag = 1;
}

#ifdef DR_PARITY_HOOKS  // only the literal (unfolded) order of FeatureNet's stage-3 head uses this kernel
// ------------------------------------------------------------------ FeatureNet skip connection
// inter = nearest_up2(coarser) + conv1x1(x) + bias   (module.py:518-531: `F.interpolate(..., scale_factor=2) + skip(...)`).
// Cin = 8 or 16 inputs, 32 outputs per pixel: 512-1024 flop against 128 B written, 128 B re-read -- a streaming
// operation.  On the MFMA convolution kernel (tile staging through LDS, K = 8 padded to a 16-wide chunk) it ran at
// 2.3 TB/s; here one lane owns 4 output channels of a pixel (8 lanes = one 128-byte pixel record, a wave = 1 KiB of
// contiguous output), its 4 x Cin weights live in registers, and the products are accumulated in the order the MFMA
// kernel accumulated them (k = 4g + s: s outer, g inner), so the result is the same fmaf chain.
template <int CIN>
__global__ __launch_bounds__(256) void k_skip_up(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                 const float *__restrict__ coarse, float *__restrict__ out, int V, int H, int W) {
  constexpr int CO = 32;
  const int q = threadIdx.x & 7;  // output channels 4q .. 4q+3
  float wr[4][CIN];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < CIN; ++c) wr[r][c] = w[(4 * q + r) * CIN + c];
  const float4 b = *reinterpret_cast<const float4 *>(bias + 4 * q);
  const size_t npix = (size_t)V * H * W, stride = (size_t)gridDim.x * (blockDim.x >> 3);
  const int Hc = H >> 1, Wc = W >> 1;
  for (size_t p = (size_t)blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3); p < npix; p += stride) {
    const int xx = (int)(p % W), yy = (int)((p / W) % H), v = (int)(p / ((size_t)W * H));
    float xi[CIN];
#pragma unroll
    for (int c = 0; c < CIN; c += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(x + p * CIN + c);
      xi[c] = t.x; xi[c + 1] = t.y; xi[c + 2] = t.z; xi[c + 3] = t.w;
    }
    const float4 up = *reinterpret_cast<const float4 *>(coarse + (((size_t)v * Hc + (yy >> 1)) * Wc + (xx >> 1)) * CO + 4 * q);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int g = 0; g < CIN / 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(wr[r][4 * g + s], xi[4 * g + s], acc[r]);
    float4 o;
    o.x = (acc[0] + b.x) + up.x; o.y = (acc[1] + b.y) + up.y; o.z = (acc[2] + b.z) + up.z; o.w = (acc[3] + b.w) + up.w;
    *reinterpret_cast<float4 *>(out + p * CO + 4 * q) = o;
  }
}

#endif  // DR_PARITY_HOOKS

// ------------------------------------------------------------------ depth hypotheses
struct PlaneArgs {
  const float *prev;  // previous stage depth (hp x wp) or nullptr for the uniform stage
  int hp, wp;
  int D;
  float dmin, interval;          // stage 1: d_k = dmin + interval * k            (module.py:1494-1496)
  float half_range, full_range;  // later : lo = max(cur - half_range, 1e-3), hi = lo + full_range
};

// F.interpolate(scale 2, bilinear, align_corners=False): src = 0.5*(dst+0.5)-0.5 clamped at 0  (cva_mvsnet.py:144-147)
__device__ inline float up2_bilinear(const float *__restrict__ prev, int hp, int wp, int y, int x) {
  const float sy = fmaxf(0.5f * ((float)y + 0.5f) - 0.5f, 0.f);
  const float sx = fmaxf(0.5f * ((float)x + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < hp - 1 ? 1 : 0), x1 = x0 + (x0 < wp - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float v00 = prev[y0 * wp + x0], v01 = prev[y0 * wp + x1], v10 = prev[y1 * wp + x0], v11 = prev[y1 * wp + x1];
  return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

struct PixelPlanes {  // per-pixel hypothesis generator
  float lo, rng, invD;
  bool uniform;
  __device__ inline float at(const PlaneArgs &p, int k) const {
    return uniform ? p.dmin + p.interval * (float)k : lo + rng * ((float)k * invD);
  }
};

__device__ inline PixelPlanes make_planes(const PlaneArgs &p, int y, int x) {
  PixelPlanes r;
  r.uniform = p.prev == nullptr;
  r.lo = 0.f; r.rng = 0.f; r.invD = 1.f / (float)p.D;
  if (!r.uniform) {
    const float cur = up2_bilinear(p.prev, p.hp, p.wp, y, x);
    r.lo = fmaxf(cur - p.half_range, 0.001f);      // module.py:1518-1523
    const float hi = r.lo + p.full_range;          // module.py:1526
    r.rng = hi - r.lo;                             // module.py:1531 (depth_max - depth_min)
  }
  return r;
}

// The hypothesis of plane k spelled with an explicit fused multiply-add (PixelPlanes::at leaves the contraction to the compiler, which
// decides differently from one kernel to the next: kernels that must agree bit for bit use this form).
__device__ inline float plane_depth(const PixelPlanes &pp, const PlaneArgs &p, int k) {
  return pp.uniform ? __builtin_fmaf(p.interval, (float)k, p.dmin) : __builtin_fmaf(pp.rng, (float)k * pp.invD, pp.lo);
}

// ------------------------------------------------------------------ cost volume
struct CostVolArgs {
  const float *feat;  // (V,h,w,C) channels-last, view 0 = reference
  const float *vfeat[kMaxSrc + 1];  // k_costvol5: the (bordered) feature map of every view of the window by pointer -- feat + v * plane, or, with the key-frame
                                    // feature cache (dr_mvsnet.hip), the cache entry that holds the image's features
  float *vol;         // (D,h,w,C)
  PlaneArgs planes;
  int V, h, w, dchunk;
  float M[kMaxSrc][12];  // per source view: rows of [rot | trans] of ref-pixel -> src-pixel (module.py:795-809)
  float gw[32];          // gate conv weight per channel
  float gA1, gB1, gA2, gB2;  // folded gate affine: g = relu(A2*relu(A1*s + B1) + B2)
  float nsrc_f;          // float(V-1)
  int view_aggregation;
  int gx, gz, nwg;       // workgroup grid: x-blocks per row, depth chunks, total (rows = nwg / (gx * gz))
  int fpad;              // k_costvol2: the feature maps carry a zero border of this many pixels (1)
  int abl;               // ablations for measurements (parity build only, DR_CV5_ABL): 1 = no gathers, 2 = no stores, 4 = no arithmetic on the taps
  int split;             // 16: the 32-channel volume of stage 1 is stored as TWO (D,h,w,16) halves one after the other (channels 0-15 | 16-31), so
                         // that each 16-channel pass of conv0 reads whole 64-byte records instead of half of every 128-byte one; 0: (D,h,w,C)
};
// float index of channel ch (a multiple of 4) of voxel `vox` = (d * h + y) * w + x in the volume
template <int C>
__device__ __forceinline__ size_t cv_vol_index(const CostVolArgs &a, size_t vox, int ch) {
  if constexpr (C == 32) {
    if (a.split) return (size_t)(ch >> 4) * ((size_t)a.planes.D * a.h * a.w * 16) + vox * 16 + (ch & 15);
  }
  return vox * C + ch;
}

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

#ifdef DR_PARITY_HOOKS  // round 2's generation (DR_COSTVOL_V1), kept for the parity build
// One lane owns CPL channels of one pixel (4: the C / 4 lanes of a pixel read its whole channels-last record with one
// instruction -- the kernel is bound by the L1's one tag look-up per cycle, PMC: 0.8 line accesses per cycle per CU with
// CPL = 8) and walks the depth planes of its chunk.
// Per (plane, view): 3 FMAs + one v_rcp (1 ulp; the reference divides, the coordinate differs by <1e-4 px) give the
// source position; taps are fetched branch-free (clamped address, zeroed weight == grid_sample's zero padding).
template <int C, int CPL>               // CPL = channels per lane (4 or 8)
__global__ __launch_bounds__(256) void k_costvol(const CostVolArgs a) {
  constexpr int NV = CPL / 4;           // float4 per lane
  constexpr int LPV = C / CPL;          // lanes per pixel
  constexpr int PXB = 256 / LPV;        // pixels per block
  const int tid = threadIdx.x, q = tid % LPV;
  // XCD-aware workgroup order.  Workgroups are dealt round-robin to the 8 XCDs (own 4 MiB L2 each); taken in launch
  // order every XCD would gather from ALL rows of all source views (tens of MB: PMC showed 0.6 / 1.7 / 1.0 GB of L2
  // misses per launch at stages 1 / 2 / 3 against 17 / 34 / 69 MB of feature maps).  Here XCD k walks the k-th band of
  // rows, row by row, x-block by x-block, depth chunk innermost, so the few hundred workgroups it has in flight share
  // a footprint of a few rows of each source view.
  const int per = (a.nwg + 7) >> 3;
  const int nid = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if (nid >= a.nwg) return;
  const int bz = nid % a.gz, bxy = nid / a.gz;
  const int x = (bxy % a.gx) * PXB + tid / LPV, y = bxy / a.gx;
  const int d0 = bz * a.dchunk, d1 = min(a.planes.D, d0 + a.dchunk);
  const bool live = x < a.w;
  const int xc = live ? x : a.w - 1;  // keep dead lanes running for the cross-lane gate sum
  const int h = a.h, w = a.w, nsrc = a.V - 1;
  const unsigned plane = (unsigned)h * w * C;  // floats per view (< 2^31 bytes for every supported size)
  const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);

  float4 ref[NV], gw[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ref[i] = ld4(a.feat + ((size_t)y * w + xc) * C + q * CPL + 4 * i);
    gw[i] = make_float4(a.gw[q * CPL + 4 * i], a.gw[q * CPL + 4 * i + 1], a.gw[q * CPL + 4 * i + 2], a.gw[q * CPL + 4 * i + 3]);
  }
  const PixelPlanes pp = make_planes(a.planes, y, xc);
  float rx[kMaxSrc], ry[kMaxSrc], rz[kMaxSrc];
#pragma unroll
  for (int v = 0; v < kMaxSrc; ++v) {
    if (v < nsrc) {
      const float *m = a.M[v];
      rx[v] = m[0] * (float)xc + m[1] * (float)y + m[2];
      ry[v] = m[4] * (float)xc + m[5] * (float)y + m[6];
      rz[v] = m[8] * (float)xc + m[9] * (float)y + m[10];
    }
  }
  const float inv_n = a.view_aggregation ? a.nsrc_f : a.nsrc_f + 1.f;

  for (int d = d0; d < d1; ++d) {
    const float depth = pp.at(a.planes, d);
    float4 acc[NV], s1[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      acc[i] = a.view_aggregation ? make_float4(0.f, 0.f, 0.f, 0.f)
                                  : make_float4(ref[i].x * ref[i].x, ref[i].y * ref[i].y, ref[i].z * ref[i].z, ref[i].w * ref[i].w);
      s1[i] = ref[i];
    }
#pragma unroll
    for (int v = 0; v < kMaxSrc; ++v) {
      if (v >= nsrc) break;
      const float *m = a.M[v];
      const float px = rx[v] * depth + m[3], py = ry[v] * depth + m[7], pz = rz[v] * depth + m[11];
      const float rcp = __builtin_amdgcn_rcpf(pz);
      const float u = px * rcp, vv = py * rcp;
      // module.py:861,887 (z < 1e-3 -> 0) and grid_sample zero padding; NaN coordinates fail `inside` too
      const bool inside = pz >= 0.001f && u > -1.f && u < (float)w && vv > -1.f && vv < (float)h;
      const float uc = inside ? u : 0.f, vc = inside ? vv : 0.f;
      const float fx0 = floorf(uc), fy0 = floorf(vc);
      const float ax = uc - fx0, ay = vc - fy0;
      const bool xl = fx0 >= 0.f, xr = fx0 < wm1, yt = fy0 >= 0.f, yb = fy0 < hm1;
      const float bx = 1.f - ax, by = 1.f - ay;
      const float w00 = (inside && xl && yt) ? bx * by : 0.f, w01 = (inside && xr && yt) ? ax * by : 0.f;
      const float w10 = (inside && xl && yb) ? bx * ay : 0.f, w11 = (inside && xr && yb) ? ax * ay : 0.f;
      const int x0 = max((int)fx0, 0), y0 = max((int)fy0, 0);
      const int x1 = min((int)fx0 + 1, w - 1), y1 = min((int)fy0 + 1, h - 1);
      const float *f = a.feat + (size_t)(v + 1) * plane + q * CPL;
      const unsigned o00 = ((unsigned)y0 * w + x0) * C, o01 = ((unsigned)y0 * w + x1) * C;
      const unsigned o10 = ((unsigned)y1 * w + x0) * C, o11 = ((unsigned)y1 * w + x1) * C;
      float s = 0.f;
      float4 d2[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 t00 = ld4(f + o00 + 4 * i), t01 = ld4(f + o01 + 4 * i), t10 = ld4(f + o10 + 4 * i), t11 = ld4(f + o11 + 4 * i);
        float4 wv;
        wv.x = t00.x * w00 + t01.x * w01 + t10.x * w10 + t11.x * w11;
        wv.y = t00.y * w00 + t01.y * w01 + t10.y * w10 + t11.y * w11;
        wv.z = t00.z * w00 + t01.z * w01 + t10.z * w10 + t11.z * w11;
        wv.w = t00.w * w00 + t01.w * w01 + t10.w * w10 + t11.w * w11;
        if (a.view_aggregation) {
          const float4 df = make_float4(wv.x - ref[i].x, wv.y - ref[i].y, wv.z - ref[i].z, wv.w - ref[i].w);
          d2[i] = make_float4(df.x * df.x, df.y * df.y, df.z * df.z, df.w * df.w);
          s += gw[i].x * d2[i].x + gw[i].y * d2[i].y + gw[i].z * d2[i].z + gw[i].w * d2[i].w;
        } else {  // plain variance incl. the reference view (module.py:1074-1075,1094-1096,1110)
          s1[i].x += wv.x; s1[i].y += wv.y; s1[i].z += wv.z; s1[i].w += wv.w;
          acc[i].x += wv.x * wv.x; acc[i].y += wv.y * wv.y; acc[i].z += wv.z * wv.z; acc[i].w += wv.w * wv.w;
        }
      }
      if (a.view_aggregation) {
#pragma unroll
        for (int msk = 1; msk < LPV; msk <<= 1) s += __shfl_xor(s, msk);
        const float g1 = fmaxf(a.gA1 * s + a.gB1, 0.f);
        const float g = fmaxf(a.gA2 * g1 + a.gB2, 0.f) + 1.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) { acc[i].x += g * d2[i].x; acc[i].y += g * d2[i].y; acc[i].z += g * d2[i].z; acc[i].w += g * d2[i].w; }
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float4 o = make_float4(acc[i].x / inv_n, acc[i].y / inv_n, acc[i].z / inv_n, acc[i].w / inv_n);
      if (!a.view_aggregation) {
        const float4 mu = make_float4(s1[i].x / inv_n, s1[i].y / inv_n, s1[i].z / inv_n, s1[i].w / inv_n);
        o = make_float4(o.x - mu.x * mu.x, o.y - mu.y * mu.y, o.z - mu.z * mu.z, o.w - mu.w * mu.w);
      }
      if (live) *reinterpret_cast<float4 *>(a.vol + cv_vol_index<C>(a, (size_t)d * h * w + (size_t)y * w + x, q * CPL + 4 * i)) = o;
    }
  }
}

#endif  // DR_PARITY_HOOKS

// k_costvol2: the same arithmetic on feature maps stored with a ONE-PIXEL ZERO BORDER ((h+2) x (w+2) per view), as a
// software-pipelined flat loop over (plane, view).
//  * Zero border: grid_sample's zero padding needs no per-tap logic.  A sample that is outside altogether (or behind the
//    camera) is moved to (-1, -1), where tap 00 is the border's zero with weight 1 and the other taps have weight 0; a
//    sample whose footprint is partly outside reads border zeros for the missing taps.  Products and their order are
//    k_costvol's (0 * w instead of t * 0 for a missing tap).  Gone per (plane, view): four tap-validity flags and weight
//    selects, the x1 / y1 clamps and three of the four address computations (one VGPR offset against two per-view
//    scalar bases: rows y0 and y0 + 1; +C floats for x0 + 1).
//  * Pipelining: k_costvol issued the four gathers of a (plane, view) and waited for them before it touched the next
//    view -- with four waves per SIMD each iteration exposed most of an L2 round trip (PMC round 2: 541 cycles per
//    wave-iteration against ~210 cycles of instruction issue).  Here the gathers of iteration i + 1 are issued before the
//    arithmetic of iteration i (two tap sets in registers, used alternately), across view and plane boundaries.
//  * The view index is a run-time loop variable (its matrix comes from scalar loads), the per-view ray coefficients are
//    recomputed (6 FMAs) instead of held in 21 registers; the gate's cross-lane channel sum uses DPP, not ds_bpermute.
struct CvTaps {
  float4 t00, t01, t10, t11;
  float w00, w01, w10, w11;
};
__device__ inline float cv_dpp_add(float s, int ctrl) {  // s + s[dpp permutation of the row]
  if (ctrl == 0) return s + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if (ctrl == 1) return s + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  return s + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));                // row_half_mirror: lane i <-> 7 - i
}
// The arithmetic of one (plane, view) sample, shared by k_costvol2 and k_costvol3 with every multiply-add spelled out: the two
// kernels then agree bit for bit whatever the compiler would have contracted in either context.
struct CvProj { int o; float w00, w01, w10, w11; int ix, iy, inside; };  // (ix, iy) = the sample's upper-left tap, -1 .. w-1 / h-1
struct CvRay { float rx, ry, rz; };  // M[:3,:3] * (x, y, 1): the part of a sample's projection that does not depend on the plane (module.py:820-822)
__device__ __forceinline__ CvRay cv_ray(const float *m, float xf, float yf) {
  CvRay R;
  R.rx = __builtin_fmaf(m[0], xf, __builtin_fmaf(m[1], yf, m[2])); R.ry = __builtin_fmaf(m[4], xf, __builtin_fmaf(m[5], yf, m[6]));
  R.rz = __builtin_fmaf(m[8], xf, __builtin_fmaf(m[9], yf, m[10]));
  return R;
}
__device__ __forceinline__ CvProj cv_project_ray(const float *m, const CvRay &R, float depth, float fw, float fh, int wp, int C);
__device__ __forceinline__ CvProj cv_project(const float *m, float depth, float xf, float yf, float fw, float fh, int wp, int C) {
  return cv_project_ray(m, cv_ray(m, xf, yf), depth, fw, fh, wp, C);
}
__device__ __forceinline__ CvProj cv_project_ray(const float *m, const CvRay &R, float depth, float fw, float fh, int wp, int C) {
  const float px = __builtin_fmaf(R.rx, depth, m[3]), py = __builtin_fmaf(R.ry, depth, m[7]), pz = __builtin_fmaf(R.rz, depth, m[11]);
  const float rcp = __builtin_amdgcn_rcpf(pz);
  const float u = px * rcp, vv = py * rcp;
  const bool inside = pz >= 0.001f && u > -1.f && u < fw && vv > -1.f && vv < fh;  // module.py:861,887; NaN fails too
  const float uc = inside ? u : -1.f, vc = inside ? vv : -1.f;
  const float fx0 = floorf(uc), fy0 = floorf(vc);
  const float ax = uc - fx0, ay = vc - fy0, bx = 1.f - ax, by = 1.f - ay;
  CvProj r;
  r.w00 = bx * by; r.w01 = ax * by; r.w10 = bx * ay; r.w11 = ax * ay;
  r.ix = (int)fx0; r.iy = (int)fy0; r.inside = inside ? 1 : 0;
  r.o = (r.iy * wp + r.ix) * C;
  return r;
}
__device__ __forceinline__ float cv_tap4(float t00, float t01, float t10, float t11, const CvTaps &T) {
  return __builtin_fmaf(t11, T.w11, __builtin_fmaf(t10, T.w10, __builtin_fmaf(t01, T.w01, t00 * T.w00)));
}
__device__ __forceinline__ float4 cv_warp(const CvTaps &T) {
  return make_float4(cv_tap4(T.t00.x, T.t01.x, T.t10.x, T.t11.x, T), cv_tap4(T.t00.y, T.t01.y, T.t10.y, T.t11.y, T),
                     cv_tap4(T.t00.z, T.t01.z, T.t10.z, T.t11.z, T), cv_tap4(T.t00.w, T.t01.w, T.t10.w, T.t11.w, T));
}
__device__ __forceinline__ float cv_gate_dot(const float4 &gw, const float4 &d2) {
  return __builtin_fmaf(gw.w, d2.w, __builtin_fmaf(gw.z, d2.z, __builtin_fmaf(gw.y, d2.y, gw.x * d2.x)));
}
template <int C>
__global__ __launch_bounds__(256) void k_costvol2(const CostVolArgs a) {
  constexpr int LPV = C / 4;            // lanes per pixel, 4 channels each
  constexpr int PXB = 256 / LPV;        // pixels per block
  const int tid = threadIdx.x, q = tid % LPV;
  const int per = (a.nwg + 7) >> 3;     // XCD-aware order, as k_costvol
  const int nid = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if (nid >= a.nwg) return;
  const int bz = nid % a.gz, bxy = nid / a.gz;
  const int x = (bxy % a.gx) * PXB + tid / LPV, y = bxy / a.gx;
  const int d0 = bz * a.dchunk, d1 = min(a.planes.D, d0 + a.dchunk);
  const bool live = x < a.w;
  const int xc = live ? x : a.w - 1;    // dead lanes keep running for the cross-lane gate sum
  const int h = a.h, w = a.w, nsrc = a.V - 1, wp = w + 2;
  const size_t vplane = (size_t)(h + 2) * wp * C;  // floats per padded view
  const float fw = (float)w, fh = (float)h, xf = (float)xc, yf = (float)y;

  const float4 ref = ld4(a.feat + ((size_t)(y + 1) * wp + xc + 1) * C + q * 4);
  const float4 gw = make_float4(a.gw[q * 4], a.gw[q * 4 + 1], a.gw[q * 4 + 2], a.gw[q * 4 + 3]);
  const PixelPlanes pp = make_planes(a.planes, y, xc);
  const float inv_n = a.view_aggregation ? a.nsrc_f : a.nsrc_f + 1.f;
  const float rcp_n = 1.f / inv_n;  // the mean over the views is taken by this factor: four IEEE divisions per voxel (40 instructions) otherwise; <= 1 ulp from x / n
  const float *f00 = a.feat + ((size_t)wp + 1) * C + q * 4;  // pixel (0, 0) of view 0, this lane's channels

  auto issue = [&](int d, int v, CvTaps &T) {
    const CvProj P = cv_project(a.M[v] /* wave-uniform: scalar loads */, plane_depth(pp, a.planes, d), xf, yf, fw, fh, wp, C);
    T.w00 = P.w00; T.w01 = P.w01; T.w10 = P.w10; T.w11 = P.w11;
    const int o = P.o;
    const float *r0 = f00 + (size_t)(v + 1) * vplane, *r1 = r0 + (size_t)wp * C;  // wave-uniform bases: rows y0 and y0 + 1
    T.t00 = ld4(r0 + o); T.t01 = ld4(r0 + o + C); T.t10 = ld4(r1 + o); T.t11 = ld4(r1 + o + C);
  };
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), s1 = acc;
  auto consume = [&](int d, int v, const CvTaps &T) {
    if (v == 0) {
      acc = a.view_aggregation ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(ref.x * ref.x, ref.y * ref.y, ref.z * ref.z, ref.w * ref.w);
      s1 = ref;
    }
    const float4 wv = cv_warp(T);
    if (a.view_aggregation) {
      const float4 df = make_float4(wv.x - ref.x, wv.y - ref.y, wv.z - ref.z, wv.w - ref.w);
      const float4 d2 = make_float4(df.x * df.x, df.y * df.y, df.z * df.z, df.w * df.w);
      float s = cv_gate_dot(gw, d2);
      if constexpr (LPV >= 2) s = cv_dpp_add(s, 0);
      if constexpr (LPV >= 4) s = cv_dpp_add(s, 1);
      if constexpr (LPV >= 8) s = cv_dpp_add(s, 2);
      const float g1 = fmaxf(__builtin_fmaf(a.gA1, s, a.gB1), 0.f);
      const float g = fmaxf(__builtin_fmaf(a.gA2, g1, a.gB2), 0.f) + 1.f;
      acc.x = __builtin_fmaf(g, d2.x, acc.x); acc.y = __builtin_fmaf(g, d2.y, acc.y); acc.z = __builtin_fmaf(g, d2.z, acc.z); acc.w = __builtin_fmaf(g, d2.w, acc.w);
    } else {  // plain variance incl. the reference view (module.py:1074-1075,1094-1096,1110)
      s1.x += wv.x; s1.y += wv.y; s1.z += wv.z; s1.w += wv.w;
      acc.x = __builtin_fmaf(wv.x, wv.x, acc.x); acc.y = __builtin_fmaf(wv.y, wv.y, acc.y); acc.z = __builtin_fmaf(wv.z, wv.z, acc.z); acc.w = __builtin_fmaf(wv.w, wv.w, acc.w);
    }
    if (v == nsrc - 1) {
      float4 o4 = make_float4(acc.x * rcp_n, acc.y * rcp_n, acc.z * rcp_n, acc.w * rcp_n);
      if (!a.view_aggregation) {
        const float4 mu = make_float4(s1.x * rcp_n, s1.y * rcp_n, s1.z * rcp_n, s1.w * rcp_n);
        o4 = make_float4(__builtin_fmaf(-mu.x, mu.x, o4.x), __builtin_fmaf(-mu.y, mu.y, o4.y), __builtin_fmaf(-mu.z, mu.z, o4.z), __builtin_fmaf(-mu.w, mu.w, o4.w));
      }
      if (live) *reinterpret_cast<float4 *>(a.vol + cv_vol_index<C>(a, (size_t)d * h * w + (size_t)y * w + x, q * 4)) = o4;
    }
  };
  const int n = (d1 - d0) * nsrc;
  if (n <= 0) return;
  CvTaps A, B;
  int d = d0, v = 0;
  issue(d, v, A);
  for (int it = 0; it < n; it += 2) {  // two iterations per trip: the tap sets A and B alternate without copies
    int dn = d, vn = v + 1;
    if (vn == nsrc) { vn = 0; ++dn; }
    issue(dn, vn, B);  // unconditional (past the end: a harmless extra gather): a branch here makes hipcc wait for ALL loads at the join
    consume(d, v, A);
    if (it + 1 >= n) break;
    d = dn; v = vn;
    dn = d; vn = v + 1;
    if (vn == nsrc) { vn = 0; ++dn; }
    issue(dn, vn, A);
    consume(d, v, B);
    d = dn; v = vn;
  }
}

// k_costvol3: k_costvol2 with the per-sample set-up SHARED by the lanes of a pixel.  In k_costvol2 the C / 4 lanes that hold a
// pixel's channels all compute the same homography, inside test, bilinear weights and tap offset for every (plane, view) --
// 45 of the ~95 vector instructions of an iteration, and vector-ALU issue is what bounds the kernel (87 M wave-instructions
// at stage 2 = 0.14 of its 0.16 ms).  Here the lanes of a pixel take DIFFERENT iterations: lane q sets up iteration
// base + (q mod LPB) of a batch of LPB = 4 (2 for C = 8) consecutive iterations, and the five results (tap offset, four
// weights) reach the other lanes through DPP quad permutations when their iteration comes up.  Per iteration the set-up is
// then 45 / LPB + 5 instructions.  Products, sums and their order are k_costvol2's: the volume is bit-identical.
// Requires (d1 - d0) * nsrc to be a multiple of LPB for every depth chunk (dchunk and D multiples of 4: the host checks).
__device__ __forceinline__ int cv_bcast_i(int x, int lpb, int j) {  // value of lane j of this lane's pixel group (lpb = 4: the quad; 2: the pair)
  if (lpb == 4) {
    switch (j) {
      case 0: return __builtin_amdgcn_mov_dpp(x, 0x00, 0xF, 0xF, true);
      case 1: return __builtin_amdgcn_mov_dpp(x, 0x55, 0xF, 0xF, true);
      case 2: return __builtin_amdgcn_mov_dpp(x, 0xAA, 0xF, 0xF, true);
      default: return __builtin_amdgcn_mov_dpp(x, 0xFF, 0xF, 0xF, true);
    }
  }
  return j == 0 ? __builtin_amdgcn_mov_dpp(x, 0xA0, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(x, 0xF5, 0xF, 0xF, true);  // [0,0,2,2] / [1,1,3,3]
}
__device__ __forceinline__ float cv_bcast_f(float x, int lpb, int j) { return __builtin_bit_cast(float, cv_bcast_i(__builtin_bit_cast(int, x), lpb, j)); }
// Round 4 tried to take a pixel's right-hand taps from the NEIGHBOURING pixel's lanes (DPP) where they are the same addresses, sending
// the now redundant loads to one shared line: 0.116 / 0.158 / 0.106 ms per stage against 0.105 / 0.149 / 0.100 -- slower, removed
// (profiles/r04_experiments.txt, 6).
#ifndef DR_CV3_MIN_WAVES  // A/B hook: minimum waves per SIMD hipcc must leave room for (register cap)
#define DR_CV3_MIN_WAVES 1
#endif
template <int C>
__global__ __launch_bounds__(256, DR_CV3_MIN_WAVES) void k_costvol3(const CostVolArgs a) {
  constexpr int LPV = C / 4;            // lanes per pixel, 4 channels each
  constexpr int PXB = 256 / LPV;        // pixels per block
  constexpr int LPB = LPV >= 4 ? 4 : 2; // iterations per batch = lanes of a pixel (within one quad) that share their set-up
  __shared__ float sM[kMaxSrc * 12];    // the views' matrices: a lane needs the one of ITS iteration's view (not wave-uniform any more)
  const int tid = threadIdx.x, q = tid % LPV, qb = q & (LPB - 1);
  for (int i = tid; i < kMaxSrc * 12; i += 256) sM[i] = a.M[i / 12][i % 12];
  __syncthreads();
  const int per = (a.nwg + 7) >> 3;     // XCD-aware order, as k_costvol
  const int nid = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if (nid >= a.nwg) return;
  const int bz = nid % a.gz, bxy = nid / a.gz;
  const int x = (bxy % a.gx) * PXB + tid / LPV, y = bxy / a.gx;
  const int d0 = bz * a.dchunk, d1 = min(a.planes.D, d0 + a.dchunk);
  const bool live = x < a.w;
  const int xc = live ? x : a.w - 1;    // dead lanes keep running for the cross-lane gate sum
  const int h = a.h, w = a.w, nsrc = a.V - 1, wp = w + 2;
  const size_t vplane = (size_t)(h + 2) * wp * C;  // floats per padded view
  const float fw = (float)w, fh = (float)h, xf = (float)xc, yf = (float)y;

  const float4 ref = ld4(a.feat + ((size_t)(y + 1) * wp + xc + 1) * C + q * 4);
  const float4 gw = make_float4(a.gw[q * 4], a.gw[q * 4 + 1], a.gw[q * 4 + 2], a.gw[q * 4 + 3]);
  const PixelPlanes pp = make_planes(a.planes, y, xc);
  const float inv_n = a.view_aggregation ? a.nsrc_f : a.nsrc_f + 1.f;
  const float rcp_n = 1.f / inv_n;
  const float *f00 = a.feat + ((size_t)wp + 1) * C + q * 4;  // pixel (0, 0) of view 0, this lane's channels

  auto project = [&](int d, int v) {  // k_costvol2's `issue` up to the tap offset and weights, for this lane's own (plane, view)
    return cv_project(sM + 12 * v, plane_depth(pp, a.planes, d), xf, yf, fw, fh, wp, C);
  };
  auto gather = [&](const CvProj &P, int j, int v, CvTaps &T) {  // iteration j of the batch: lane j's set-up, view v (uniform)
    const int o = cv_bcast_i(P.o, LPB, j);
    T.w00 = cv_bcast_f(P.w00, LPB, j); T.w01 = cv_bcast_f(P.w01, LPB, j); T.w10 = cv_bcast_f(P.w10, LPB, j); T.w11 = cv_bcast_f(P.w11, LPB, j);
    const float *r0 = f00 + (size_t)(v + 1) * vplane, *r1 = r0 + (size_t)wp * C;  // wave-uniform bases: rows y0 and y0 + 1
    T.t00 = ld4(r0 + o); T.t01 = ld4(r0 + o + C); T.t10 = ld4(r1 + o); T.t11 = ld4(r1 + o + C);
  };
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), s1 = acc;
  auto consume = [&](int d, int v, const CvTaps &T) {
    if (v == 0) {
      acc = a.view_aggregation ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(ref.x * ref.x, ref.y * ref.y, ref.z * ref.z, ref.w * ref.w);
      s1 = ref;
    }
    const float4 wv = cv_warp(T);
    if (a.view_aggregation) {
      const float4 df = make_float4(wv.x - ref.x, wv.y - ref.y, wv.z - ref.z, wv.w - ref.w);
      const float4 d2 = make_float4(df.x * df.x, df.y * df.y, df.z * df.z, df.w * df.w);
      float s = cv_gate_dot(gw, d2);
      if constexpr (LPV >= 2) s = cv_dpp_add(s, 0);
      if constexpr (LPV >= 4) s = cv_dpp_add(s, 1);
      if constexpr (LPV >= 8) s = cv_dpp_add(s, 2);
      const float g1 = fmaxf(__builtin_fmaf(a.gA1, s, a.gB1), 0.f);
      const float g = fmaxf(__builtin_fmaf(a.gA2, g1, a.gB2), 0.f) + 1.f;
      acc.x = __builtin_fmaf(g, d2.x, acc.x); acc.y = __builtin_fmaf(g, d2.y, acc.y); acc.z = __builtin_fmaf(g, d2.z, acc.z); acc.w = __builtin_fmaf(g, d2.w, acc.w);
    } else {  // plain variance incl. the reference view (module.py:1074-1075,1094-1096,1110)
      s1.x += wv.x; s1.y += wv.y; s1.z += wv.z; s1.w += wv.w;
      acc.x = __builtin_fmaf(wv.x, wv.x, acc.x); acc.y = __builtin_fmaf(wv.y, wv.y, acc.y); acc.z = __builtin_fmaf(wv.z, wv.z, acc.z); acc.w = __builtin_fmaf(wv.w, wv.w, acc.w);
    }
    if (v == nsrc - 1) {
      float4 o4 = make_float4(acc.x * rcp_n, acc.y * rcp_n, acc.z * rcp_n, acc.w * rcp_n);
      if (!a.view_aggregation) {
        const float4 mu = make_float4(s1.x * rcp_n, s1.y * rcp_n, s1.z * rcp_n, s1.w * rcp_n);
        o4 = make_float4(__builtin_fmaf(-mu.x, mu.x, o4.x), __builtin_fmaf(-mu.y, mu.y, o4.y), __builtin_fmaf(-mu.z, mu.z, o4.z), __builtin_fmaf(-mu.w, mu.w, o4.w));
      }
      if (live) *reinterpret_cast<float4 *>(a.vol + cv_vol_index<C>(a, (size_t)d * h * w + (size_t)y * w + x, q * 4)) = o4;
    }
  };
  const int n = (d1 - d0) * nsrc;  // a multiple of LPB (host)
  if (n <= 0) return;
  // this lane's own iteration of the current batch, advanced by LPB per batch
  int dq = d0, vq = qb;
  while (vq >= nsrc) { vq -= nsrc; ++dq; }
  auto advance_own = [&]() { vq += LPB; while (vq >= nsrc) { vq -= nsrc; ++dq; } };
  CvProj P = project(dq, vq);
  advance_own();
  CvTaps TA, TB;
  int d = d0, v = 0;        // the iteration being consumed (uniform)
  int vn = 0;               // view of the iteration whose taps are being gathered (uniform), one ahead of v
  gather(P, 0, vn, TA);
  for (int base = 0; base < n; base += LPB) {
    const CvProj Pn = project(dq, vq);  // the next batch's set-up (past the end in the last batch: a harmless extra sample)
    advance_own();
#pragma unroll
    for (int j = 0; j < LPB; ++j) {
      if (++vn == nsrc) vn = 0;
      CvTaps &cur = (j & 1) ? TB : TA, &nxt = (j & 1) ? TA : TB;
      if (j + 1 < LPB) gather(P, j + 1, vn, nxt);
      else gather(Pn, 0, vn, nxt);  // LPB is even: iteration 0 of a batch always lands in TA
      consume(d, v, cur);
      if (++v == nsrc) { v = 0; ++d; }
    }
    P = Pn;
  }
}

// k_costvol5 (round 6): the sweep in VIEW-OUTER / PLANE-INNER order.  k_costvol2/3 walk (plane outer, view inner) with one accumulator: every
// iteration changes the view, so the view's matrix, its row bases and the pixel's ray M[:3,:3](x, y, 1) are per-iteration work, the loop is a
// run-time (plane, view) counter with a reset and a store hidden behind branches, and consecutive gathers of a lane land in six different
// images.  Here a workgroup takes its DCH planes of ONE view after the other with DCH float4 accumulators in registers:
//   * per voxel the views are still added in the order v = 0 .. nsrc - 1 with the same products (cv_project_ray / cv_warp / the gate are
//     k_costvol3's, every multiply-add spelled out), so the volume is BIT-IDENTICAL to k_costvol3's;
//   * the view's matrix is wave-uniform again (scalar loads from the kernel arguments, no LDS copy), the ray is computed once per view
//     (6 of the set-up's FMAs leave the plane loop), the plane loop is fully unrolled (no counters, no reset / store branches), and the four
//     gathers are one 32-bit byte offset against two scalar row bases with immediate offsets 0 / 4 C (no 64-bit vector address arithmetic);
//   * the lanes of a pixel still share the set-up as in k_costvol3: lane q of a batch of LPB planes projects plane q, the tap offset and
//     the four weights travel by DPP;
//   * consecutive iterations of a lane walk ONE epipolar line, plane by plane: at the headline shape and depth range 66 / 39 / 13 % of the
//     steps (stage 1 / 2 / 3) land in the SAME 2 x 2 footprint and 17 / 24 / 14 % one texel beside it (tools/study/costvol_footprint_stat.py),
//     so a sample's lines are in the L1 from the plane before.
// View-aggregation models, depth chunks of exactly DCH planes; everything else stays on k_costvol3 / k_costvol2.
template <int C, int DCH, int REUSE, int ROWS>
__global__ __launch_bounds__(256) void k_costvol5(const CostVolArgs a) {
  constexpr int LPV = C / 4;            // lanes per pixel, 4 channels each
  constexpr int PXB = 256 / LPV;        // pixels per block
  constexpr int LPB = LPV >= 4 ? 4 : 2; // planes per batch = lanes of a pixel (within one quad) that share their set-up
  constexpr int NB = DCH / LPB;         // batches per view
  static_assert(DCH % LPB == 0 && DCH % 2 == 0, "whole batches, and every view starts in tap set A");
  const int tid = threadIdx.x, q = tid % LPV, qb = q & (LPB - 1);
  const int per = (a.nwg + 7) >> 3;     // XCD-aware order, as k_costvol
  const int nid = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if (nid >= a.nwg) return;
  const int bz = nid % a.gz, bxy = nid / a.gz;
  // ROWS = 4: the workgroup's four waves take four consecutive rows of one x segment (a.gx segments of PXB / 4 pixels) -- the footprints of rows y and
  // y + 1 share a source row, so the workgroup touches 5 source rows where four row segments side by side touch 8
  const int x = ROWS == 4 ? (bxy % a.gx) * (PXB / 4) + (tid & 63) / LPV : (bxy % a.gx) * PXB + tid / LPV, y = ROWS == 4 ? (bxy / a.gx) * 4 + (tid >> 6) : bxy / a.gx;
  const int d0 = bz * DCH;
  const bool live = x < a.w && y < a.h;
  const int xc = min(x, a.w - 1), yc = min(y, a.h - 1);    // dead lanes keep running for the cross-lane gate sum
  const int h = a.h, w = a.w, nsrc = a.V - 1, wp = w + 2;
  const float fw = (float)w, fh = (float)h, xf = (float)xc, yf = (float)yc;
  if (nsrc <= 0) return;

  const float4 ref = ld4(a.vfeat[0] + ((size_t)(yc + 1) * wp + xc + 1) * C + q * 4);
  const float4 gw = make_float4(a.gw[q * 4], a.gw[q * 4 + 1], a.gw[q * 4 + 2], a.gw[q * 4 + 3]);
  const PixelPlanes pp = make_planes(a.planes, yc, xc);
  const float rcp_n = 1.f / a.nsrc_f;
  float dep[NB];                        // this lane's own plane of every batch
#pragma unroll
  for (int b = 0; b < NB; ++b) dep[b] = plane_depth(pp, a.planes, d0 + b * LPB + qb);
  // byte offset of a sample's upper-left tap from pixel (-1, -1) of the padded view (never negative: ix, iy >= -1), this lane's channels included
  const unsigned obias = (unsigned)((wp + 1) * C + q * 4) * 4u;
  const size_t rowbytes = (size_t)wp * C * 4;

  unsigned ob_prev = 0;
  // plane j of the batch: lane j's set-up, view v (uniform).  REUSE: where the sample's footprint is the one of the lane's previous plane (`prev`, same
  // view), its four taps are already in registers: the lane issues no gather (no L1 request) and copies them.
  auto gather = [&](const CvProj &P, int j, int v, CvTaps &T, const CvTaps &prev, bool first) {
    const unsigned ob = (unsigned)cv_bcast_i(P.o, LPB, j) * 4u + obias;
    T.w00 = cv_bcast_f(P.w00, LPB, j); T.w01 = cv_bcast_f(P.w01, LPB, j); T.w10 = cv_bcast_f(P.w10, LPB, j); T.w11 = cv_bcast_f(P.w11, LPB, j);
    const char *r0 = reinterpret_cast<const char *>(a.vfeat[v + 1]), *r1 = r0 + rowbytes;  // wave-uniform bases (a scalar load from the kernel arguments): rows y0 and y0 + 1
    if (REUSE && !first && ob == ob_prev) {
      T.t00 = prev.t00; T.t01 = prev.t01; T.t10 = prev.t10; T.t11 = prev.t11;
#ifdef DR_PARITY_HOOKS
    } else if (a.abl & 1) {
      T.t00 = T.t01 = T.t10 = T.t11 = make_float4(T.w00, T.w01, T.w10, T.w11);
#endif
    } else {
      T.t00 = ld4(reinterpret_cast<const float *>(r0 + ob)); T.t01 = ld4(reinterpret_cast<const float *>(r0 + ob) + C);
      T.t10 = ld4(reinterpret_cast<const float *>(r1 + ob)); T.t11 = ld4(reinterpret_cast<const float *>(r1 + ob) + C);
    }
    if (REUSE) ob_prev = ob;
  };
  float4 acc[DCH];
#pragma unroll
  for (int j = 0; j < DCH; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto consume = [&](float4 &ac, const CvTaps &T) {
#ifdef DR_PARITY_HOOKS
    if (a.abl & 4) { ac.x += T.t00.x + T.t01.y; ac.y += T.t10.z; ac.z += T.t11.w; ac.w += T.w00; return; }
#endif
    const float4 wv = cv_warp(T);
    const float4 df = make_float4(wv.x - ref.x, wv.y - ref.y, wv.z - ref.z, wv.w - ref.w);
    const float4 d2 = make_float4(df.x * df.x, df.y * df.y, df.z * df.z, df.w * df.w);
    float s = cv_gate_dot(gw, d2);
    if constexpr (LPV >= 2) s = cv_dpp_add(s, 0);
    if constexpr (LPV >= 4) s = cv_dpp_add(s, 1);
    if constexpr (LPV >= 8) s = cv_dpp_add(s, 2);
    const float g1 = fmaxf(__builtin_fmaf(a.gA1, s, a.gB1), 0.f);
    const float g = fmaxf(__builtin_fmaf(a.gA2, g1, a.gB2), 0.f) + 1.f;
    ac.x = __builtin_fmaf(g, d2.x, ac.x); ac.y = __builtin_fmaf(g, d2.y, ac.y); ac.z = __builtin_fmaf(g, d2.z, ac.z); ac.w = __builtin_fmaf(g, d2.w, ac.w);
  };
  CvRay R = cv_ray(a.M[0], xf, yf);
  CvProj P = cv_project_ray(a.M[0], R, dep[0], fw, fh, wp, C);
  // REUSE: ONE tap set and no software pipelining.  A lane whose footprint repeats keeps its taps where they are (no gather, no copy); a lane whose footprint moved
  // gathers into the same registers and waits.  Against the two-set pipeline (gathers of plane j + 1 in flight under the arithmetic of plane j, reused taps COPIED from
  // one set to the other: eight moves per sample) the waves, not the lane's own pipeline, hide the latency: 0.096 / 0.140 / 0.105 -> 0.078 / 0.125 / 0.097 ms
  // (profiles/r06_costvol_ab.txt; six or seven waves per SIMD make no difference, and a register cap that spills two dwords into the view loop costs 30 %).
  // Without reuse every sample gathers, and the pipelined form below is the better one (the parity build's DR_CV5_REUSE=0).
  if constexpr (REUSE != 0) {
    CvTaps T;
    T.t00 = T.t01 = T.t10 = T.t11 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int v = 0; v < nsrc; ++v) {
      const float *m = a.M[v];
      const CvRay Rv = cv_ray(m, xf, yf);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const CvProj Pb = cv_project_ray(m, Rv, dep[b], fw, fh, wp, C);
#pragma unroll
        for (int jj = 0; jj < LPB; ++jj) {
          const unsigned ob = (unsigned)cv_bcast_i(Pb.o, LPB, jj) * 4u + obias;
          T.w00 = cv_bcast_f(Pb.w00, LPB, jj); T.w01 = cv_bcast_f(Pb.w01, LPB, jj); T.w10 = cv_bcast_f(Pb.w10, LPB, jj); T.w11 = cv_bcast_f(Pb.w11, LPB, jj);
          if ((b == 0 && jj == 0) || ob != ob_prev) {
            const char *r0 = reinterpret_cast<const char *>(a.vfeat[v + 1]), *r1 = r0 + rowbytes;
            T.t00 = ld4(reinterpret_cast<const float *>(r0 + ob)); T.t01 = ld4(reinterpret_cast<const float *>(r0 + ob) + C);
            T.t10 = ld4(reinterpret_cast<const float *>(r1 + ob)); T.t11 = ld4(reinterpret_cast<const float *>(r1 + ob) + C);
          }
          ob_prev = ob;
          consume(acc[b * LPB + jj], T);
        }
      }
    }
  } else {
  CvTaps TA, TB;
  gather(P, 0, 0, TA, TA, true);
  for (int v = 0; v < nsrc; ++v) {
    const int vn = min(v + 1, nsrc - 1);  // past the last view: a harmless extra sample of the last view's first plane
    const CvRay Rn = cv_ray(a.M[vn], xf, yf);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const CvProj Pn = b + 1 < NB ? cv_project_ray(a.M[v], R, dep[b + 1 < NB ? b + 1 : 0], fw, fh, wp, C) : cv_project_ray(a.M[vn], Rn, dep[0], fw, fh, wp, C);
#pragma unroll
      for (int jj = 0; jj < LPB; ++jj) {
        const int j = b * LPB + jj;
        CvTaps &cur = (j & 1) ? TB : TA, &nxt = (j & 1) ? TA : TB;
        if (jj + 1 < LPB) gather(P, jj + 1, v, nxt, cur, false);
        else gather(Pn, 0, b + 1 < NB ? v : vn, nxt, cur, b + 1 == NB);  // (the first plane of a view always gathers)
        consume(acc[j], cur);
      }
      P = Pn;
    }
    R = Rn;
  }
  }
#ifdef DR_PARITY_HOOKS
  if ((a.abl & 2) && acc[0].x != 123.456f) return;
#endif
  if (live) {
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      const float4 o4 = make_float4(acc[j].x * rcp_n, acc[j].y * rcp_n, acc[j].z * rcp_n, acc[j].w * rcp_n);
      *reinterpret_cast<float4 *>(a.vol + cv_vol_index<C>(a, (size_t)(d0 + j) * h * w + (size_t)y * w + x, q * 4)) = o4;
    }
  }
}

#ifdef DR_PARITY_HOOKS  // measured slower than k_costvol3 (see MvsSwitches::cv4_stages): built into the parity library only
// k_costvol4 (round 4): k_costvol3 with the source taps STAGED THROUGH LDS -- north_star's "LDS staging of per-pixel feature slices".
// What bounds k_costvol2/3 is the L1's tag path, not bytes and not ALU issue: every (pixel, plane, view) sample is four gathers of the
// pixel's whole channel record, i.e. four cache-line look-ups per sample, although the taps of neighbouring pixels and of neighbouring
// planes are the same few lines (a pixel's right tap is its neighbour's left tap; rows y0 / y0 + 1 serve two output rows; consecutive
// planes of the fine stages move the sample by a fraction of a pixel).  Here a workgroup owns a small pixel TILE (256 / (C / 4) pixels)
// and a depth chunk, and walks (view, group of LPB planes) steps: the union footprint of the tile's samples of a step -- their bounding
// box in the source view, known once the lanes have projected their own sample -- is fetched ONCE, row by row, by LDS-DMA
// (global_load_lds_dwordx4: coalesced 16-byte pieces, no staging registers) into one of two LDS buffers while the previous step is
// consumed out of the other; the four taps of a sample are then ds_read_b128.  One line look-up per staged line instead of one per tap:
// 3-4x fewer at the fine stages.  A step whose box does not fit the buffer (planes of the uniform stage that are metres apart, a strongly
// rotated view) gathers from global memory exactly as k_costvol3 does -- decided per step, uniformly for the workgroup.
// Arithmetic: cv_project / cv_warp / the gate / the accumulation are k_costvol3's, per (pixel, plane) in the same view order, on the same
// tap values: the volume is bit-identical (test_lds_staged_cost_volume_is_bit_identical).
// MEASURED (MI355X, 640 x 480 x 7, profiles/r04_experiments.txt 5): 0.121 / 0.163 / 0.105 ms per stage against k_costvol3's 0.106 / 0.150 /
// 0.099 (first version, before the per-sample instruction diet: 0.143 / 0.181 / 0.114) -- the staged form removes three quarters of the
// line look-ups but pays 27 % more vector instructions per sample, two barriers per step and an LDS-DMA latency that four samples of
// arithmetic do not cover.  Kept, with its test, in the parity build; the product runs k_costvol3 (whose neighbour-tap borrowing variant was slower too and is gone, see above).  View-aggregation models only (the plain-
// variance form needs a second accumulator set per plane; it stays on k_costvol3).
constexpr int kCv4Slots = 1024;  // float4 slots per LDS buffer (16 KiB; two buffers)
template <int C> struct Cv4Shape {
  static constexpr int LPV = C / 4, NPX = 256 / LPV, TW = C == 8 ? 16 : 8, TH = NPX / TW, LPB = LPV >= 4 ? 4 : 2;
};
// min over the wave of a (DPP inside the rows of 16, four readlanes across them); wave-uniform result
__device__ __forceinline__ int cv4_wave_min(int v) {
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true));   // row_half_mirror
  v = min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true));   // row_mirror
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// SP = planes per step (4 or 8): more planes amortise the step's fixed work (projection of the lane's own samples, the box, the
// staging requests, two barriers) but widen the box; the host picks 8 for the fine stages and 4 for the uniform stage.
template <int C, int DCH, int SP>
__global__ __launch_bounds__(256) void k_costvol4(const CostVolArgs a) {
  constexpr int LPV = Cv4Shape<C>::LPV, TW = Cv4Shape<C>::TW, TH = Cv4Shape<C>::TH, LPB = Cv4Shape<C>::LPB;
  constexpr int SPL = SP / LPB, NG = DCH / SP;  // a step = SP planes of one view; a lane projects SPL of them (the C / 4 lanes of a pixel share the rest)
  static_assert(DCH % SP == 0 && SP % LPB == 0, "depth chunks are whole steps, steps whole batches");
  __shared__ float4 buf[2][kCv4Slots];
  __shared__ int bb[2][4];            // per step parity: min ix, min iy, -max ix, -max iy over the samples that lie inside the view
  __shared__ float sM[kMaxSrc * 12];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = tid % LPV, qb = q & (LPB - 1), pix = tid / LPV;
  for (int i = tid; i < kMaxSrc * 12; i += 256) sM[i] = a.M[i / 12][i % 12];
  if (tid < 8) bb[tid >> 2][tid & 3] = INT_MAX;
  const int per = (a.nwg + 7) >> 3;   // XCD-aware order: XCD k walks the k-th band of tile rows, depth chunks of a tile innermost
  const int nid = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if (nid >= a.nwg) return;           // (uniform: before the first barrier)
  __syncthreads();
  const int bz = nid % a.gz, bxy = nid / a.gz;
  const int x = (bxy % a.gx) * TW + pix % TW, y = (bxy / a.gx) * TH + pix / TW;   // the host launches this kernel for whole tiles only
  const int d0 = bz * DCH;
  const int h = a.h, w = a.w, nsrc = a.V - 1, wp = w + 2;
  const size_t vplane = (size_t)(h + 2) * wp * C;  // floats per padded view
  const float fw = (float)w, fh = (float)h, xf = (float)x, yf = (float)y;
  const float4 ref = ld4(a.feat + ((size_t)(y + 1) * wp + x + 1) * C + q * 4);
  const float4 gw = make_float4(a.gw[q * 4], a.gw[q * 4 + 1], a.gw[q * 4 + 2], a.gw[q * 4 + 3]);
  const PixelPlanes pp = make_planes(a.planes, y, x);
  const float rcp_n = 1.f / a.nsrc_f;
  const float *f00 = a.feat + ((size_t)wp + 1) * C;  // pixel (0, 0) of view 0 (tap coordinates start at -1: the zero border)

  struct Box { int x0, y0, bw, fits; };  // wave-uniform (scalar registers): origin and width of the staged box; fits = 0: this step gathers from global memory
  // A lane's own sample of a step, as the consumers need it: where the upper-left tap lies and the four weights.  A sample outside the
  // view gets four ZERO weights (k_costvol3 gives it weight 1 on the border's zero): the warped value is then +-0 instead of +0, and
  // only its squared difference to the reference feature is ever used -- the same bits.
  struct Samp { int o, ix, iy, off; float w00, w01, w10, w11; };
  struct Proj { Samp s[SPL]; };
  auto project = [&](int v, int g) {  // planes d0 + g * SP + b * LPB + qb
    Proj P;
#pragma unroll
    for (int b = 0; b < SPL; ++b) {
      const CvProj c = cv_project(sM + 12 * v, plane_depth(pp, a.planes, d0 + g * SP + b * LPB + qb), xf, yf, fw, fh, wp, C);
      Samp &t = P.s[b];
      t.o = c.o; t.off = 0;
      t.ix = c.inside ? c.ix : INT_MAX; t.iy = c.iy;
      t.w00 = c.inside ? c.w00 : 0.f; t.w01 = c.inside ? c.w01 : 0.f; t.w10 = c.inside ? c.w10 : 0.f; t.w11 = c.inside ? c.w11 : 0.f;
    }
    return P;
  };
  auto post_box = [&](const Proj &P, int par) {  // min / max of the inside samples' tap coordinates: the wave's, then (LDS atomics) the workgroup's
    int mnx = INT_MAX, mny = INT_MAX, nmx = INT_MAX, nmy = INT_MAX;  // (the maxima as minima of the negated coordinate)
#pragma unroll
    for (int b = 0; b < SPL; ++b)
      if (P.s[b].ix != INT_MAX) { mnx = min(mnx, P.s[b].ix); mny = min(mny, P.s[b].iy); nmx = min(nmx, -P.s[b].ix); nmy = min(nmy, -P.s[b].iy); }
    mnx = cv4_wave_min(mnx); mny = cv4_wave_min(mny); nmx = cv4_wave_min(nmx); nmy = cv4_wave_min(nmy);
    if (lane == 0) { atomicMin(&bb[par][0], mnx); atomicMin(&bb[par][1], mny); atomicMin(&bb[par][2], nmx); atomicMin(&bb[par][3], nmy); }
  };
  // the agreed box of a step -> origin / width, the staging of its rows as 16-byte pieces (element e = row r, float4 c of the row), and
  // every lane's own tap offset inside the box (float4 slots, without its channel quad)
  auto read_and_stage = [&](Proj &P, int v, int par) {
    Box b;
    const int x0 = __builtin_amdgcn_readfirstlane(bb[par][0]), y0 = __builtin_amdgcn_readfirstlane(bb[par][1]);
    const int nx1 = __builtin_amdgcn_readfirstlane(bb[par][2]), ny1 = __builtin_amdgcn_readfirstlane(bb[par][3]);
    const bool any = x0 != INT_MAX;
    const int bw = any ? -nx1 - x0 + 2 : 1, bh = any ? -ny1 - y0 + 2 : 1;  // + the right / lower tap
    b.x0 = any ? x0 : 0; b.y0 = any ? y0 : 0; b.bw = bw;
    b.fits = bw * bh * LPV <= kCv4Slots ? 1 : 0;  // (no sample inside the view: nothing to stage -- every weight is zero, any slot will do)
#pragma unroll
    for (int s = 0; s < SPL; ++s) P.s[s].off = P.s[s].ix != INT_MAX ? ((P.s[s].iy - b.y0) * bw + (P.s[s].ix - b.x0)) * LPV : 0;
    if (any && b.fits) {
      const int rowlen = bw * LPV, total = rowlen * bh;
      const float inv = 1.0f / (float)rowlen;
      const float *src0 = f00 + (size_t)(v + 1) * vplane + ((ptrdiff_t)y0 * wp + x0) * C;  // (tap coordinates start at -1: still inside the padded view)
      for (int p0 = wave * 64; p0 < total; p0 += 256) {
        const int e = p0 + lane;
        int r = (int)((float)e * inv);
        if (r * rowlen > e) --r;
        if ((r + 1) * rowlen <= e) ++r;
        const int c = e - r * rowlen;
        const float *src = e < total ? src0 + ((size_t)r * wp * C + (size_t)c * 4) : a.feat;  // (a.feat: 16 zero bytes of the border; never read back)
        conv_a_dma16(src, __builtin_amdgcn_readfirstlane(conv_a_lds_addr(&buf[par][p0])));
      }
    }
    return b;
  };
  // Steps run plane group by plane group, the views of a group innermost (per (pixel, plane) the views are still accumulated in their
  // order): the accumulators of a step are the SP planes of ONE group -- SP registers, written out when the group's last view is done.
  float4 acc[SP];
#pragma unroll
  for (int j = 0; j < SP; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto consume = [&](const Proj &PP, const Box &b, int v, int par) {
    const float *r0 = f00 + q * 4 + (size_t)(v + 1) * vplane, *r1 = r0 + (size_t)wp * C;  // the global fall-back's row bases (k_costvol3's)
    const float4 *tile = &buf[par][q];
    const int row = b.bw * LPV;
#pragma unroll
    for (int bi = 0; bi < SPL; ++bi)
#pragma unroll
    for (int j = 0; j < LPB; ++j) {
      const Samp &P = PP.s[bi];
      CvTaps T;
      T.w00 = cv_bcast_f(P.w00, LPB, j); T.w01 = cv_bcast_f(P.w01, LPB, j); T.w10 = cv_bcast_f(P.w10, LPB, j); T.w11 = cv_bcast_f(P.w11, LPB, j);
      if (b.fits) {  // (uniform) taps out of the staged box
        const float4 *t = tile + cv_bcast_i(P.off, LPB, j);
        T.t00 = t[0]; T.t01 = t[LPV]; T.t10 = t[row]; T.t11 = t[row + LPV];
      } else {
        const int o = cv_bcast_i(P.o, LPB, j);
        T.t00 = ld4(r0 + o); T.t01 = ld4(r0 + o + C); T.t10 = ld4(r1 + o); T.t11 = ld4(r1 + o + C);
      }
      const float4 wv = cv_warp(T);
      const float4 df = make_float4(wv.x - ref.x, wv.y - ref.y, wv.z - ref.z, wv.w - ref.w);
      const float4 d2 = make_float4(df.x * df.x, df.y * df.y, df.z * df.z, df.w * df.w);
      float sg = cv_gate_dot(gw, d2);
      if constexpr (LPV >= 2) sg = cv_dpp_add(sg, 0);
      if constexpr (LPV >= 4) sg = cv_dpp_add(sg, 1);
      if constexpr (LPV >= 8) sg = cv_dpp_add(sg, 2);
      const float g1 = fmaxf(__builtin_fmaf(a.gA1, sg, a.gB1), 0.f);
      const float gg = fmaxf(__builtin_fmaf(a.gA2, g1, a.gB2), 0.f) + 1.f;
      float4 &A = acc[bi * LPB + j];
      A.x = __builtin_fmaf(gg, d2.x, A.x); A.y = __builtin_fmaf(gg, d2.y, A.y); A.z = __builtin_fmaf(gg, d2.z, A.z); A.w = __builtin_fmaf(gg, d2.w, A.w);
    }
  };
  const int S = NG * nsrc;  // step s = (group s / nsrc, view s % nsrc)
  if (S > 0) {
    // prologue: step 0 projected, boxed and staged
    Proj P = project(0, 0);
    post_box(P, 0);
    __syncthreads();
    Box B = read_and_stage(P, 0, 0);
    conv_a_wait_dma();
    __syncthreads();
    int g = 0, v = 0;
    for (int s = 0; s < S; ++s) {
      // the NEXT step's samples are projected and its box is agreed on and requested while this step's taps are read
      const int par = s & 1;
      const bool last = s == S - 1;
      const int vn = v + 1 < nsrc ? v + 1 : 0, gn = v + 1 < nsrc ? g : g + 1;
      Proj Pn = P;
      if (!last) { Pn = project(vn, gn); post_box(Pn, par ^ 1); }
      __syncthreads();
      Box Bn = B;
      if (!last) Bn = read_and_stage(Pn, vn, par ^ 1);
      consume(P, B, v, par);
      if (v == nsrc - 1) {  // the group's planes are complete
#pragma unroll
        for (int j = 0; j < SP; ++j) {
          const float4 o4 = make_float4(acc[j].x * rcp_n, acc[j].y * rcp_n, acc[j].z * rcp_n, acc[j].w * rcp_n);
          *reinterpret_cast<float4 *>(a.vol + cv_vol_index<C>(a, (size_t)(d0 + g * SP + j) * h * w + (size_t)y * w + x, q * 4)) = o4;
          acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      conv_a_wait_dma();
      if (tid < 4) bb[par][tid] = INT_MAX;  // this parity's box was read a barrier ago; it is posted again after the next one
      __syncthreads();
      P = Pn; B = Bn; v = vn; g = gn;
    }
  } else {  // (no source view on this rank: the host zeroes the volume instead of launching; kept for completeness)
    for (int i = 0; i < DCH; ++i)
      *reinterpret_cast<float4 *>(a.vol + cv_vol_index<C>(a, (size_t)(d0 + i) * h * w + (size_t)y * w + x, q * 4)) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

#endif  // DR_PARITY_HOOKS

// ------------------------------------------------------------------ prob conv (Cout = 1)
// CostRegNet.prob = Conv3d(8, 1, 3, padding=1, bias=False) (module.py:575): 216 MACs per voxel and a single output
// channel -- no matrix shape to speak of, so it runs on the vector pipe: one lane = 4 consecutive x outputs,
// weights are wave-uniform (scalar loads), inputs are float4 channels-last reads served by L1.
// Each lane owns the column (y, x0..x0+3) of a z-chunk and MARCHES along z: one input plane (3 rows x 6 positions
// x 8 channels) is loaded once and feeds the three output planes it touches, so L1 traffic is a third of a
// plane-at-a-time stencil.
// Workgroup order (gz > 0): as in k_costvol, XCD k (= blockIdx.x % 8, own L2) walks the k-th band of rows, z chunks of a
// pixel block innermost -- the three input rows an output row needs and the two halo planes of a z chunk are then
// fetched into ONE L2 (launch order spread them over all eight: PMC showed 208 MB fetched per launch against 62 MB of input).
#ifdef DR_PARITY_HOOKS  // round 2's generation (DR_PROB_V1), kept for the parity build
template <int XO>  // x outputs per lane (4: fewest L1 accesses per output; 2: twice the waves to hide their latency)
__global__ __launch_bounds__(256) void k_prob(const float *__restrict__ x, const float *__restrict__ wt /*[27][8]*/,
                                              float *__restrict__ out, int D, int h, int w, int zchunk, int gz, int nwg) {
  const int wq = w / XO;
  int bx = blockIdx.x, bz = blockIdx.y;
  if (gz > 0) {
    const int per = (nwg + 7) >> 3, nid = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (nid >= nwg) return;
    bz = nid % gz; bx = nid / gz;
  }
  const int n = bx * blockDim.x + threadIdx.x;
  if (n >= h * wq) return;
  const int xq = n % wq, y = n / wq, x0 = xq * XO;
  const int z0 = bz * zchunk, z1 = min(D, z0 + zchunk);
  // acc[j][o]: output plane (zz - 1 + j) while input plane zz is being consumed
  float a0[XO], a1[XO], a2[XO];
#pragma unroll
  for (int o = 0; o < XO; ++o) a0[o] = a1[o] = a2[o] = 0.f;
  for (int zz = z0 - 1; zz <= z1; ++zz) {
    if (zz >= 0 && zz < D) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int yy = y + kh - 1;
        if (yy < 0 || yy >= h) continue;
        const float *row = x + ((size_t)zz * h + yy) * w * 8;
        float4 lo[XO + 2], hi[XO + 2];
#pragma unroll
        for (int i = 0; i < XO + 2; ++i) {
          const int xx = x0 - 1 + i;
          if (xx >= 0 && xx < w) { lo[i] = ld4(row + (size_t)xx * 8); hi[i] = ld4(row + (size_t)xx * 8 + 4); }
          else { lo[i] = make_float4(0.f, 0.f, 0.f, 0.f); hi[i] = lo[i]; }
        }
#define DR_DOT8(A, I, WK) A += lo[I].x * WK[0] + lo[I].y * WK[1] + lo[I].z * WK[2] + lo[I].w * WK[3] + hi[I].x * WK[4] + hi[I].y * WK[5] + hi[I].z * WK[6] + hi[I].w * WK[7]
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          // input plane zz is tap kd = 2 of output zz-1, kd = 1 of output zz, kd = 0 of output zz+1
          const float *w2 = wt + ((2 * 3 + kh) * 3 + kw) * 8, *w1 = wt + ((1 * 3 + kh) * 3 + kw) * 8, *w0 = wt + ((0 * 3 + kh) * 3 + kw) * 8;
#pragma unroll
          for (int o = 0; o < XO; ++o) { DR_DOT8(a0[o], kw + o, w2); DR_DOT8(a1[o], kw + o, w1); DR_DOT8(a2[o], kw + o, w0); }
        }
#undef DR_DOT8
      }
    }
    const int zo = zz - 1;  // complete once input plane zz has been consumed
    if (zo >= z0 && zo < z1) {
      float *dst = out + ((size_t)zo * h + y) * w + x0;
      if (XO == 4) *reinterpret_cast<float4 *>(dst) = make_float4(a0[0], a0[1], a0[2], a0[3]);
      else if (XO == 2) *reinterpret_cast<float2 *>(dst) = make_float2(a0[0], a0[1]);
      else dst[0] = a0[0];
    }
#pragma unroll
    for (int o = 0; o < XO; ++o) { a0[o] = a1[o]; a1[o] = a2[o]; a2[o] = 0.f; }
  }
}

#endif  // DR_PARITY_HOOKS

// k_prob2: the same layer with the input plane staged through LDS.  k_prob reads every input position nine times per
// plane from L1 (three rows x three columns per lane: 288 B per output and plane against 64 B/clk/CU of L1 bandwidth,
// and PMC showed 3.3x the input bytes fetched from L2); here a workgroup of 256 lanes owns a 4-row x 64-column tile and
// marches along z: one input plane of the tile (6 x 66 positions, two float4 halves kept in separate arrays so that
// consecutive lanes read consecutive 16-byte slots) is staged ONCE per plane -- global -> registers while the previous
// plane is being consumed, registers -> the other LDS buffer afterwards, one barrier per plane -- and serves the nine
// taps of all 256 lanes and the three output planes it contributes to.  The products are k_prob's; they are summed in two
// interleaved chains (even and odd channels) instead of one.
constexpr int kProbTY = 4, kProbTX = 64, kProbPos = (kProbTY + 2) * (kProbTX + 2);  // 396 staged positions per plane (NR = 1)
// NR (round 5): rows per lane.  A lane that owns NR vertically adjacent logits reads the NR + 2 rows around them once (9 LDS reads per logit at NR = 4
// against 18), the 24 wave-uniform weights of a tap are fetched once for all of them (the scalar loads and their s_waitcnt were what the NR = 1 kernel waited
// for: 20 waits per 108 packed FMAs), and the tile's halo shrinks from 1.55 to 1.16 of its interior.  Per logit the products and their order are unchanged:
// bit-identical for every NR.  MEASURED (MI355X, gpurun call c17 of round 5): 0.019 / 0.028 / 0.026 ms per stage at NR = 1, 0.029 / 0.044 / 0.038 at NR = 2,
// 0.059 / 0.084 / 0.074 at NR = 4 -- half / a quarter of the workgroups, each with 1.7 / 3 x the LDS, lose more latency hiding than the shared reads save: the
// product runs NR = 1 (DR_PROB_ROWS selects the others for A/B).
// (the body as a device function: k_prob2_regress below runs it and the stage's regression in one launch).  Returns false for a surplus workgroup;
// (yo, xo) = the lane's first pixel.
template <int NR>
__device__ inline bool prob2_body(const float *__restrict__ x, const float *__restrict__ wt /*[27][8]*/, float *__restrict__ out, int D, int h, int w, int zchunk,
                                  int gx, int gy, int gz, int nwg, int &yo_out, int &xo_out) {
  constexpr int TY = kProbTY * NR, POS = (TY + 2) * (kProbTX + 2), NS = (POS + 255) / 256;
  extern __shared__ float4 prob_lds[];  // [buffer 2][channel half 2][POS]
  const int per = (nwg + 7) >> 3, nid = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);  // XCD k walks the k-th band of tile rows
  if (nid >= nwg) return false;
  const int bz = nid % gz, bxy = nid / gz, bx = bxy % gx, by = bxy / gx;
  const int tid = threadIdx.x, tx = tid & 63, ty = (tid >> 6) * NR;
  const int x0 = bx * kProbTX, y0 = by * TY, xo = x0 + tx, yo = y0 + ty;
  const int z0 = bz * zchunk, z1 = min(D, z0 + zchunk);
  yo_out = yo; xo_out = xo;
  // this thread's share of a plane: staged positions tid, tid + 256, ... (both halves each)
  int spos[NS];
  const float *sptr[NS];
  bool sin[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int p = tid + k * 256, py = p / (kProbTX + 2), px = p - py * (kProbTX + 2);
    const int gyy = y0 - 1 + py, gxx = x0 - 1 + px;
    spos[k] = p < POS ? p : -1;
    sin[k] = p < POS && gyy >= 0 && gyy < h && gxx >= 0 && gxx < w;
    sptr[k] = x + ((size_t)(sin[k] ? gyy : 0) * w + (sin[k] ? gxx : 0)) * 8;
  }
  const size_t plane = (size_t)h * w * 8;
  float4 r[NS][2];
  auto fetch = [&](int zz) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      r[k][0] = r[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sin[k] && zz >= 0 && zz < D) { r[k][0] = ld4(sptr[k] + (size_t)zz * plane); r[k][1] = ld4(sptr[k] + (size_t)zz * plane + 4); }
    }
  };
  auto stash = [&](int b) {
    float4 *lo = prob_lds + (size_t)b * 2 * POS, *hi = lo + POS;
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (spos[k] >= 0) { lo[spos[k]] = r[k][0]; hi[spos[k]] = r[k][1]; }
  };
  // output planes zz-1, zz, zz+1 while input plane zz is being consumed.  Each accumulator is a PAIR (even / odd channels, added at
  // the end): the 8-channel dot product of a tap is then four packed fmas (v_pk_fma_f32: two fp32 fmas per lane and instruction)
  // instead of eight scalar ones.
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0[NR], a1[NR], a2[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) a0[q] = a1[q] = a2[q] = f2{0.f, 0.f};
  fetch(z0 - 1);
  stash(0);
  int b = 0;
  for (int zz = z0 - 1; zz <= z1; ++zz, b ^= 1) {
    __syncthreads();  // plane zz is in buffer b; everybody is done with buffer b ^ 1
    if (zz + 1 <= z1) fetch(zz + 1);
    if (zz >= 0 && zz < D) {
      const float4 *lo = prob_lds + (size_t)b * 2 * POS, *hi = lo + POS;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const f2 *w2 = reinterpret_cast<const f2 *>(wt + ((2 * 3 + kh) * 3 + kw) * 8), *w1 = reinterpret_cast<const f2 *>(wt + ((1 * 3 + kh) * 3 + kw) * 8),
                   *w0 = reinterpret_cast<const f2 *>(wt + ((0 * 3 + kh) * 3 + kw) * 8);
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            if (yo + q + kh - 1 < 0 || yo + q + kh - 1 >= h) continue;  // as k_prob: rows outside the image are skipped (their staged zeros are never read)
            const int p = (ty + q + kh) * (kProbTX + 2) + tx + kw;
            const float4 l4 = lo[p], h4 = hi[p];
            const f2 x01 = {l4.x, l4.y}, x23 = {l4.z, l4.w}, x45 = {h4.x, h4.y}, x67 = {h4.z, h4.w};
#define DR_DOT8(A, WK) A = __builtin_elementwise_fma(x01, WK[0], A); A = __builtin_elementwise_fma(x23, WK[1], A); A = __builtin_elementwise_fma(x45, WK[2], A); A = __builtin_elementwise_fma(x67, WK[3], A)
            DR_DOT8(a0[q], w2); DR_DOT8(a1[q], w1); DR_DOT8(a2[q], w0);
#undef DR_DOT8
          }
        }
      }
    }
    const int zo = zz - 1;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      if (zo >= z0 && zo < z1 && xo < w && yo + q < h) out[((size_t)zo * h + yo + q) * w + xo] = a0[q].x + a0[q].y;
      a0[q] = a1[q]; a1[q] = a2[q]; a2[q] = f2{0.f, 0.f};
    }
    if (zz + 1 <= z1) stash(b ^ 1);
  }
  return true;
}
template <int NR>
__global__ __launch_bounds__(256) void k_prob2(const float *__restrict__ x, const float *__restrict__ wt /*[27][8]*/,
                                               float *__restrict__ out, int D, int h, int w, int zchunk, int gx, int gy, int gz, int nwg) {
  int yo, xo;
  (void)prob2_body<NR>(x, wt, out, D, h, w, zchunk, gx, gy, gz, nwg, yo, xo);
}
inline size_t prob2_lds_bytes(int NR) { return (size_t)2 * 2 * (kProbTY * NR + 2) * (kProbTX + 2) * sizeof(float4); }

// ------------------------------------------------------------------ folded out.stage3: border term
// out.stage3(up(inter2) + skip3(c3)) is linear, so it is evaluated as conv3x3(c3; Wout . Wskip) + conv3x3(up(inter2); Wout) + B with
// B[co] = sum over the 9 taps of T[tap][co], T[tap][co] = sum_c Wout[co][c][tap] * bskip[c].  B is exact for interior pixels; at the
// image border the taps that fall outside see the ZERO padding of inter3, not the bias, so their T is taken out again here
// (one lane per (border pixel, channel); 2 (H + W) - 4 pixels per view).
__global__ __launch_bounds__(256) void k_out3_border(float *__restrict__ out, const float *__restrict__ T /*[9][8]*/, int V, int H, int W, int rowstride,
                                                     size_t planestride) {
  const int per = 2 * (H + W) - 4, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= V * per * 8) return;
  const int co = n & 7, b = (n >> 3) % per, v = (n >> 3) / per;
  int y, x;
  if (b < W) { y = 0; x = b; }
  else if (b < 2 * W) { y = H - 1; x = b - W; }
  else if (b < 2 * W + H - 2) { y = b - 2 * W + 1; x = 0; }
  else { y = b - (2 * W + H - 2) + 1; x = W - 1; }
  float corr = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xx = x + kx - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) corr += T[(ky * 3 + kx) * 8 + co];
    }
  float *p = out + (size_t)v * planestride + (size_t)y * rowstride + (size_t)x * 8 + co;
  *p -= corr;
}

// ------------------------------------------------------------------ regression
// expf whose result cannot be fused into its consumer: with -ffp-contract=fast hipcc folds the exponential's last multiply into
// `sum += ...` where the value has a single use and cannot where it is reused, so two spellings of the same regression would
// differ in the last bit.  Both kernels below use this form and are bit-identical to each other.
__device__ inline float expf_value(float x) {
  float r = expf(x);
  asm volatile("" : "+v"(r));
  return r;
}
// (plane depth and the two running sums are spelled with explicit fused multiply-adds, see plane_depth)
struct RegressArgs {
  const float *logits;  // (D,h,w)
  float *depth, *conf;  // (h,w)
  PlaneArgs planes;
  int h, w;
};

__global__ __launch_bounds__(256) void k_regress(const RegressArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, hw = a.h * a.w;
  if (n >= hw) return;
  const int y = n / a.w, x = n - y * a.w, D = a.planes.D;
  const PixelPlanes pp = make_planes(a.planes, y, x);
  float mx = -INFINITY;
  for (int k = 0; k < D; ++k) mx = fmaxf(mx, a.logits[(size_t)k * hw + n]);
  float sum = 0.f;
  for (int k = 0; k < D; ++k) sum += expf_value(a.logits[(size_t)k * hw + n] - mx);
  float dep = 0.f, ek = 0.f;
  for (int k = 0; k < D; ++k) {
    const float p = expf_value(a.logits[(size_t)k * hw + n] - mx) / sum;
    dep = __builtin_fmaf(p, plane_depth(pp, a.planes, k), dep);
    ek = __builtin_fmaf(p, (float)k, ek);
  }
  int idx = (int)ek;  // .long() truncation, module.py:1131
  idx = min(max(idx, 0), D - 1);
  float c = 0.f;
  for (int k = idx - 1; k <= idx + 2; ++k)
    if (k >= 0 && k < D) c += expf_value(a.logits[(size_t)k * hw + n] - mx) / sum;
  a.depth[n] = dep;
  a.conf[n] = c;
}

// The same regression with the pixel's D logits held in registers (D known at compile time: the plane counts the models use):
// one round of loads, issued back to back, instead of three dependent passes over global memory plus the confidence window.
// The arithmetic and its order are k_regress's.
template <int D>
__device__ inline void regress_regs(float (&v)[D], const RegressArgs &a, int y, int x, int n) {
  const PixelPlanes pp = make_planes(a.planes, y, x);
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < D; ++k) mx = fmaxf(mx, v[k]);
#pragma unroll
  for (int k = 0; k < D; ++k) v[k] = expf_value(v[k] - mx);  // (the three-pass kernel recomputes this value in every pass)
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) sum += v[k];
  float dep = 0.f, ek = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const float p = v[k] / sum;
    dep = __builtin_fmaf(p, plane_depth(pp, a.planes, k), dep);
    ek = __builtin_fmaf(p, (float)k, ek);
  }
  int idx = (int)ek;  // .long() truncation, module.py:1131
  idx = min(max(idx, 0), D - 1);
  float c = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k)  // the four planes idx-1 .. idx+2, in ascending order as k_regress adds them
    if (k >= idx - 1 && k <= idx + 2) c += v[k] / sum;
  a.depth[n] = dep;
  a.conf[n] = c;
}
template <int D>
__global__ __launch_bounds__(256) void k_regress_r(const RegressArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, hw = a.h * a.w;
  if (n >= hw) return;
  const int y = n / a.w, x = n - y * a.w;
  float v[D];
#pragma unroll
  for (int k = 0; k < D; ++k) v[k] = a.logits[(size_t)k * hw + n];
  regress_regs<D>(v, a, y, x, n);
}
// prob + regression in one launch (round 5; stage 3 of the headline configuration: its D = 8 planes are one depth chunk of k_prob2 already, so the
// lane that owns a pixel has written all RD logits of it itself): k_prob2's march unchanged, then the lane reads its own logits back (a thread
// observes its own stores; they come from L2) and k_regress_r's arithmetic follows -- the same expressions on the same values, bit-identical to
// the two launches.  (Keeping the logits in registers instead needs the march unrolled RD + 2 times; hipcc then stops unrolling the tap loops.)
template <int RD>
__global__ __launch_bounds__(256) void k_prob2_regress(const float *__restrict__ x, const float *__restrict__ wt /*[27][8]*/, float *out, int h, int w,
                                                       int gx, int gy, int nwg, const RegressArgs rg) {
  int yo, xo;
  if (!prob2_body<1>(x, wt, out, RD, h, w, RD, gx, gy, 1, nwg, yo, xo)) return;
  if (yo >= h || xo >= w) return;
  const int n = yo * w + xo;
  const size_t hw = (size_t)h * w;
  float v[RD];
#pragma unroll
  for (int k = 0; k < RD; ++k) v[k] = out[(size_t)k * hw + n];
  regress_regs<RD>(v, rg, yo, xo, n);
}

// ------------------------------------------------------------------ edge filter
// edge(x,y) = 15th smallest of |d(nb) - d(c)| over the zero-padded 5x5 window (incl. the centre's 0).
// (It also resets the radix select's state for this depth map -- [0] prefix value, [1] prefix mask, [2] the rank looked for, [3] threshold
// bits -- which used to be a launch of its own.  Round 4 also moved each level's scan into the histogram kernel's last workgroup (ticket +
// fence): 9 -> 5 launches, bit-identical, but 0.070 ms against 0.058 for the filter -- the scan behind a device-wide fence and a ticket round
// trip costs more than the 6 us launch it saves; removed.)
__device__ inline void edge_pixel(const float *__restrict__ depth, float *__restrict__ edge, int h, int w, int n) {
  const int y = n / w, x = n - y * w;
  const float c = depth[n];
  float v[25];
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int yy = y + dy, xx = x + dx;
      const float nb = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? depth[yy * w + xx] : 0.f;
      v[(dy + 2) * 5 + dx + 2] = fabsf(nb - c);
    }
#ifdef DR_EDGE_RANK_COUNT  // round 1-2: rank by counting, 1250 compares per pixel (A/B build)
  float kth = 0.f;
#pragma unroll
  for (int i = 0; i < 25; ++i) {
    int less = 0, leq = 0;
#pragma unroll
    for (int jx = 0; jx < 25; ++jx) {
      less += v[jx] < v[i] ? 1 : 0;
      leq += v[jx] <= v[i] ? 1 : 0;
    }
    if (less <= 14 && leq > 14) kth = v[i];  // k = 15 (1-based), module.py:1336,1343
  }
  edge[n] = kth;
#else
  // the same order statistic from a sorting network: Knuth's merge exchange (TAOCP 5.2.2 M) for 25 inputs has 138 comparators,
  // the 113 below are the ones element 14 of the sorted order depends on (of which the compiler drops the unused min or max
  // halves).  A k-th smallest VALUE does not depend on how it is found, so the result is the counting form's bit for bit.
#define CE(A, B) { const float lo_ = fminf(v[A], v[B]), hi_ = fmaxf(v[A], v[B]); v[A] = lo_; v[B] = hi_; }
  CE(0,16) CE(1,17) CE(2,18) CE(3,19) CE(4,20) CE(5,21) CE(6,22) CE(7,23) CE(8,24) CE(0,8) CE(1,9) CE(2,10)
  CE(3,11) CE(4,12) CE(5,13) CE(6,14) CE(7,15) CE(16,24) CE(8,16) CE(9,17) CE(10,18) CE(11,19) CE(12,20) CE(13,21)
  CE(14,22) CE(15,23) CE(0,4) CE(1,5) CE(2,6) CE(3,7) CE(8,12) CE(9,13) CE(10,14) CE(11,15) CE(16,20) CE(17,21)
  CE(18,22) CE(19,23) CE(4,16) CE(5,17) CE(6,18) CE(7,19) CE(12,24) CE(4,8) CE(5,9) CE(6,10) CE(7,11) CE(12,16)
  CE(13,17) CE(14,18) CE(15,19) CE(20,24) CE(0,2) CE(1,3) CE(4,6) CE(5,7) CE(8,10) CE(9,11) CE(12,14) CE(13,15)
  CE(16,18) CE(17,19) CE(20,22) CE(21,23) CE(2,16) CE(3,17) CE(6,20) CE(7,21) CE(10,24) CE(2,8) CE(3,9) CE(6,12)
  CE(7,13) CE(10,16) CE(11,17) CE(14,20) CE(15,21) CE(18,24) CE(2,4) CE(3,5) CE(6,8) CE(7,9) CE(10,12) CE(11,13)
  CE(14,16) CE(15,17) CE(18,20) CE(19,21) CE(22,24) CE(0,1) CE(2,3) CE(4,5) CE(6,7) CE(8,9) CE(10,11) CE(12,13)
  CE(14,15) CE(16,17) CE(18,19) CE(20,21) CE(22,23) CE(1,16) CE(3,18) CE(5,20) CE(7,22) CE(9,24) CE(7,14) CE(9,16)
  CE(11,18) CE(13,20) CE(11,14) CE(13,16) CE(13,14)
#undef CE
  edge[n] = v[14];  // k = 15 (1-based), module.py:1336,1343
#endif
}
__global__ __launch_bounds__(256) void k_edge(const float *__restrict__ depth, float *__restrict__ edge, int h, int w, unsigned *__restrict__ state, unsigned rank) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < 4) state[n] = n == 2 ? rank : 0u;
  if (n >= h * w) return;
  edge_pixel(depth, edge, h, w, n);
}
__global__ __launch_bounds__(256) void k_edge2(const float *__restrict__ depth, float *__restrict__ edge, int h, int w, unsigned *__restrict__ state, unsigned *__restrict__ hist,
                                               unsigned rank) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < 4) state[n] = n == 2 ? rank : 0u;
  for (int i = n; i < 3 * 2048; i += gridDim.x * blockDim.x) hist[i] = 0u;
  if (n >= h * w) return;
  edge_pixel(depth, edge, h, w, n);
}

// Exact k-th smallest of non-negative floats by a 3-level radix select on the bit pattern
// (monotone for x >= 0).  state: [0] prefix value, [1] prefix mask, [2] remaining rank, [3] threshold bits.
__global__ __launch_bounds__(256) void k_hist(const float *__restrict__ edge, int n, int shift, int bits,
                                              const unsigned *__restrict__ state, unsigned *__restrict__ hist) {
  __shared__ unsigned sh[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  const unsigned pv = state[0], pm = state[1], mask = (1u << bits) - 1u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned key = __float_as_uint(edge[i]);
    if ((key & pm) == pv) atomicAdd(&sh[(key >> shift) & mask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x)
    if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

__global__ __launch_bounds__(256) void k_scan(unsigned *__restrict__ state, unsigned *__restrict__ hist, int shift, int bits) {
  // one block of 256 threads, 8 bins each: block prefix sum, locate the bin holding rank state[2]
  __shared__ unsigned part[256];
  __shared__ unsigned sel[2];
  const int t = threadIdx.x;
  const unsigned nb = 1u << bits;
  unsigned loc[8], s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { loc[i] = hist[t * 8 + i]; s += loc[i]; }
  part[t] = s;
  if (t == 0) { sel[0] = nb - 1u; sel[1] = 0; }
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
    const unsigned add = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  const unsigned rank = state[2], excl = part[t] - s;
  if (s && excl <= rank && rank < excl + s) {
    unsigned cum = excl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (cum + loc[i] > rank) { sel[0] = t * 8 + i; sel[1] = cum; break; }
      cum += loc[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) hist[t * 8 + i] = 0;
  if (t == 0) {
    const unsigned v = state[0] | (sel[0] << shift);
    state[0] = v;
    state[1] |= ((nb - 1u) << shift);
    state[2] = rank - sel[1];
    state[3] = v;
  }
}

// ---- the same radix select with every level's scan folded into the kernel that needs its result (round 5: 8 -> 5 launches) ----
// A one-workgroup k_scan between two histogram levels costs a launch and a dependent kernel boundary for 2 us of work.  Here EVERY workgroup of
// the next kernel repeats that scan over the previous level's 2048 bins (8 KB from L2) before it starts -- no ticket, no device-wide fence (round
// 4's last-workgroup form paid more for those than the launch it saved): the kernel boundary that is there anyway orders the histogram before
// its readers.  The select's state lives in one 4-word slot per level -- slot L is written by workgroup 0 of the kernel that finished level L - 1 and
// is only read by later kernels -- and every level has its own histogram (zeroed by k_edge2, which runs before all of them).
//   state slot: [0] prefix value, [1] prefix mask, [2] remaining rank, [3] threshold bits (complete after the last level)
struct SelectState { unsigned pv, pm, rank, thr; };
// block-wide: the bin of `hist` (1 << bits bins, 2048 words allocated, zero beyond) that holds rank st.rank, appended to the prefix
__device__ inline SelectState select_scan(const unsigned *__restrict__ hist, const SelectState st, int shift, int bits, unsigned *part /*[256]*/, unsigned *sel /*[2]*/) {
  const int t = threadIdx.x;
  const unsigned nb = 1u << bits;
  unsigned loc[8], s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { loc[i] = hist[t * 8 + i]; s += loc[i]; }
  part[t] = s;
  if (t == 0) { sel[0] = nb - 1u; sel[1] = 0; }
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan (k_scan's)
    const unsigned add = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  const unsigned excl = part[t] - s;
  if (s && excl <= st.rank && st.rank < excl + s) {
    unsigned cum = excl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (cum + loc[i] > st.rank) { sel[0] = t * 8 + i; sel[1] = cum; break; }
      cum += loc[i];
    }
  }
  __syncthreads();
  SelectState r;
  r.pv = st.pv | (sel[0] << shift);
  r.pm = st.pm | ((nb - 1u) << shift);
  r.rank = st.rank - sel[1];
  r.thr = r.pv;
  __syncthreads();  // (part / sel may be reused by the caller)
  return r;
}
// (k_edge2 above = k_edge + the three zeroed histograms; level 0 is k_hist on slot 0 and the first histogram)
// histogram of level `level` (1 or 2: level 0 is k_hist on slot 0); prologue: the scan of level - 1
__global__ __launch_bounds__(256) void k_hist_s(const float *__restrict__ edge, int n, int shift_prev, int bits_prev, int shift, int bits, int level,
                                                unsigned *__restrict__ state, unsigned *__restrict__ hist) {
  __shared__ unsigned sh[2048];
  __shared__ unsigned part[256], sel[2];
  const SelectState *slots = reinterpret_cast<const SelectState *>(state);
  const SelectState st = select_scan(hist + (level - 1) * 2048, slots[level - 1], shift_prev, bits_prev, part, sel);
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<SelectState *>(state)[level] = st;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  const unsigned mask = (1u << bits) - 1u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned key = __float_as_uint(edge[i]);
    if ((key & st.pm) == st.pv) atomicAdd(&sh[(key >> shift) & mask], 1u);
  }
  __syncthreads();
  unsigned *out = hist + level * 2048;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x)
    if (sh[i]) atomicAdd(&out[i], sh[i]);
}
// k_apply with the last level's scan as its prologue
__global__ __launch_bounds__(256) void k_apply_s(const float *__restrict__ edge, unsigned *__restrict__ state, const unsigned *__restrict__ hist, int shift_last, int bits_last,
                                                 const float *__restrict__ depth_dense, const float *__restrict__ conf_dense,
                                                 float *__restrict__ depth, float *__restrict__ conf, int n) {
  __shared__ unsigned part[256], sel[2];
  const SelectState st = select_scan(hist + 2 * 2048, reinterpret_cast<const SelectState *>(state)[2], shift_last, bits_last, part, sel);
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<SelectState *>(state)[3] = st;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float thr = __uint_as_float(st.thr);
  const bool m = edge[i] > thr;  // strict, module.py:1357
  depth[i] = m ? 0.f : depth_dense[i];
  conf[i] = m ? 0.f : conf_dense[i];
}

__global__ __launch_bounds__(256) void k_apply(const float *__restrict__ edge, const unsigned *__restrict__ state,
                                               const float *__restrict__ depth_dense, const float *__restrict__ conf_dense,
                                               float *__restrict__ depth, float *__restrict__ conf, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float thr = __uint_as_float(state[3]);
  const bool m = edge[i] > thr;  // strict, module.py:1357
  depth[i] = m ? 0.f : depth_dense[i];
  conf[i] = m ? 0.f : conf_dense[i];
}

}  // namespace dr
