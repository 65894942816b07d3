// conv_mfma.h -- the one convolution kernel of the depth pipeline: a tap-table implicit GEMM on
// the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32, exact fp32 == an fmaf chain).
//
// It replaces every dense contraction the reference hands to cuDNN through ATen
// (FeatureNet.forward cva_mvsnet/models/module.py:496-531, CostRegNet.forward module.py:577-600):
// Conv2d 3x3 / 5x5-stride-2 / 1x1, Conv3d 3^3 (stride 1, 2, (1,2,2)) and ConvTranspose3d 3^3
// stride 2 (as ONE stride-1 conv with 8*Cout parity rows), with the eval-mode BatchNorm folded into a per-channel
// scale/bias, ReLU, the UNet residual adds and FeatureNet's nearest-upsample-add fused in the epilogue.
//
// Data layout: channels-last everywhere, tensor = (D|V, H, W, C) fp32.
//
// GEMM mapping (one wave = 64 lanes):
//   A (16 x 4)  = weights : row i = output row (channel), 4 K-values        -> lane (i = l&15, g = l>>4)
//   B (4 x 16)  = inputs  : col j = output position (16 consecutive along W) -> lane (j = l&15, g = l>>4)
//   D (16 x 16) : lane holds rows 4g..4g+3 of column j  => one float4 channels-last store per lane.
// K is the flattened (tap, cin) axis, walked in chunks of 16: each lane fetches ONE float4 (4
// consecutive cin of "its" tap) from the LDS-staged input tile and ONE float4 of pre-packed weights,
// and feeds four back-to-back MFMAs (k-order inside the dot product is free, so lane-group g owns
// K = 16q+4g .. +3).  Taps are a runtime table of LDS offsets, which is what lets the same kernel do
// strided convs, transposed convs (dense parity rows) and the two "shifted-weights" modes:
//   XPAIR: a Cout=8 layer is run as a stride-(1,1,2) conv with a 4-wide kernel and 16 output rows
//          (8 channels x 2 adjacent x) on the output viewed as (D,H,W/2,16): 75 % MFMA row use, not 50 %.
//   X8   : the Cout=1 `prob` layer is run as 8 x-shifts per column on the output viewed as (D,H,W/8,8).
#pragma once
#include <algorithm>
#include <atomic>
#include <functional>

#include "dr_common.h"

namespace dr {

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct ConvClass {  // one output-parity class of a launch (plain convs have exactly one)
  int NU;        // K chunks per channel pass
  int tap_base;  // offset into tapoff[]
  int w_base;    // offset into wpk[] (float4 units)
  int ooz, ooy, oox;
};

struct ConvArgs {
  const float *in;
  float *out;
  const float4 *wpk;
  const float *scale, *bias, *add;
  const int *tapoff;
  const ConvClass *cls;
  int inD, inH, inW, inC;
  int outD, outH, outW, outC;
  int nPD, nPH, nPW;
  int sz, sy, sx, pz, py, px;
  int omz, omy, omx;
  int TZ, TY, TXT, TZI, TYI, TXI;
  int npass, ctTot, rows_valid, relu, add_mode, addH, addW;
  int par_rows, par_map;    // transposed layers: rows per output parity (= Cout) and 3 bits (z,y,x) per parity; 0 otherwise
  unsigned magicX, magicY;  // ceil(2^32 / TXI), ceil(2^32 / TYI): exact division of tile positions (< 2^16)
  int nuMax;                // max K chunks per pass over the classes (sizes the LDS weight area)
  int tilesD, tilesH, tilesW;
};

constexpr int kConvThreads = 256;
constexpr size_t kConvMaxLds = 160 * 1024;  // gfx950: 160 KiB LDS per CU, one workgroup may take all of it

// ---- epilogue: folded BN, ReLU, residual / upsample add, one float4 (4 channels) per lane ----
template <int CT, int PT>
__device__ inline void conv_epilogue(const ConvArgs &a, const ConvClass &cls, floatx4 (&acc)[CT][PT], int wave, int j, int g, int ct0,
                                     int pz0, int py0, int px0) {
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int tau = wave * PT + pt;
    const int xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
    const int qz = pz0 + zt, qy = py0 + yt, qx = px0 + xt * 16 + j;
    if (qz >= a.nPD || qy >= a.nPH || qx >= a.nPW) continue;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c0 = (ct0 + ct) * 16 + 4 * g;
      if (c0 >= a.rows_valid) continue;
      int oz = qz * a.omz + cls.ooz, oy = qy * a.omy + cls.ooy, ox = qx * a.omx + cls.oox, ch = c0;
      if (a.par_rows) {  // transposed layer: this lane's 4 rows are 4 channels of output parity q
        const int q = c0 / a.par_rows, bits = (a.par_map >> (3 * q)) & 7;
        ch = c0 - q * a.par_rows;
        oz += (bits >> 2) & 1; oy += (bits >> 1) & 1; ox += bits & 1;
      }
      const size_t obase = (((size_t)oz * a.outH + oy) * a.outW + ox) * a.outC + ch;
      size_t abase = obase;
      if (a.add_mode == 2) abase = (((size_t)oz * a.addH + (oy >> 1)) * a.addW + (ox >> 1)) * a.outC + ch;
      const float4 sc = *reinterpret_cast<const float4 *>(a.scale + c0);
      const float4 bi = *reinterpret_cast<const float4 *>(a.bias + c0);
      float4 v;
      v.x = acc[ct][pt][0] * sc.x + bi.x;
      v.y = acc[ct][pt][1] * sc.y + bi.y;
      v.z = acc[ct][pt][2] * sc.z + bi.z;
      v.w = acc[ct][pt][3] * sc.w + bi.w;
      if (a.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (a.add_mode) {
        const float4 r = *reinterpret_cast<const float4 *>(a.add + abase);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      *reinterpret_cast<float4 *>(a.out + obase) = v;
    }
  }
}

// ---- K loop over the chunks of one channel pass, operands of chunk u+1 fetched under the MFMAs of u ----
// Two operand register sets used alternately (the loop is unrolled by two): renaming the prefetched set into the current
// one at the end of every iteration costs 8 v_mov_b64 whose results the next MFMAs have to wait for -- 118 vs 104 TFLOP/s
// (one wave per SIMD) and 138 vs 120-127 (two) in tools/ubench/mfma_lds2.hip, which isolates exactly this loop.
template <int CT, int PT>
__device__ inline void conv_chunk_mfma(const float4 (&av)[CT], const float4 (&bv)[PT], floatx4 (&acc)[CT][PT]) {
  // consecutive MFMAs go to different accumulators (16x16x4: 32-cycle issue, 40-cycle dependent latency)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct].x, bv[pt].x, acc[ct][pt], 0, 0, 0);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct].y, bv[pt].y, acc[ct][pt], 0, 0, 0);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct].z, bv[pt].z, acc[ct][pt], 0, 0, 0);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct].w, bv[pt].w, acc[ct][pt], 0, 0, 0);
}
template <int CT, int PT>
__device__ inline void conv_chunk_load(const float *lds, const float4 *wp, int toff, int u, const int (&base)[PT], float4 (&av)[CT],
                                       float4 (&bv)[PT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) av[ct] = wp[(u * CT + ct) * 64];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) bv[pt] = *reinterpret_cast<const float4 *>(lds + base[pt] + toff);
}
// Anchor prefetched operands at the END of the MFMA block before them: without a use there hipcc sinks the loads in
// front of their own MFMAs and every chunk eats a full LDS round trip.
template <int CT, int PT>
__device__ inline void conv_chunk_anchor(const float4 (&av)[CT], const float4 (&bv)[PT], int toff) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) asm volatile("" ::"v"(av[ct].x), "v"(av[ct].y), "v"(av[ct].z), "v"(av[ct].w));
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) asm volatile("" ::"v"(bv[pt].x), "v"(bv[pt].y), "v"(bv[pt].z), "v"(bv[pt].w));
  asm volatile("" ::"v"(toff));
}
template <int CT, int PT>
__device__ inline void conv_kloop(const float *lds, const float4 *wl, const int *tp, int TPC, int NU, int lane, const int (&base)[PT],
                                  floatx4 (&acc)[CT][PT]) {
  const float4 *wp = wl + lane;
  float4 a0[CT], b0[PT], a1[CT], b1[PT];
  // the tap offset of a chunk is itself an LDS read: it is fetched one chunk before the operands that need it
  int tA = tp[0], tB = tp[min(1, NU - 1) * TPC];
  conv_chunk_load<CT, PT>(lds, wp, tA, 0, base, a0, b0);
  int u = 0;
  for (; u + 1 < NU; u += 2) {
    // sched_barrier: the operand reads of the NEXT chunk are issued before the MFMAs of this one and waited for after them
    conv_chunk_load<CT, PT>(lds, wp, tB, u + 1, base, a1, b1);
    tA = tp[min(u + 2, NU - 1) * TPC];
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_mfma<CT, PT>(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_anchor<CT, PT>(a1, b1, tA);
    conv_chunk_load<CT, PT>(lds, wp, tA, min(u + 2, NU - 1), base, a0, b0);
    tB = tp[min(u + 3, NU - 1) * TPC];
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_mfma<CT, PT>(a1, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_anchor<CT, PT>(a0, b0, tB);
  }
  if (u < NU) conv_chunk_mfma<CT, PT>(a0, b0, acc);  // odd chunk count: set 0 holds the last chunk
}

// grid = (tiles, parity classes, output-row groups).  PT = position tiles (16 positions each) per wave.
template <int CI, int CT, int PT>
__global__ __launch_bounds__(kConvThreads) void k_conv(const ConvArgs a) {
  extern __shared__ float4 lds4[];
  float *lds = reinterpret_cast<float *>(lds4);
  constexpr int CIS = CI + 4;  // LDS floats per staged position (+4: spreads b128 reads over bank slots)
  constexpr int TPC = 16 / CI; // taps per 16-wide K chunk
  constexpr int C4 = CI / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const ConvClass cls = a.cls[blockIdx.y];
  const int ct0 = blockIdx.z * CT;

  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (own L2 each), so XCD k takes the k-th
  // contiguous range of tiles (x fastest, then y, then z): neighbouring tiles, which share their halo, share an L2.
  // grid.x is padded to a multiple of 8 (so that XCD == blockIdx.x % 8 for every row group); surplus workgroups exit.
  const int ntiles = a.tilesD * a.tilesH * a.tilesW, per_xcd = (ntiles + 7) >> 3;
  int b = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (b >= ntiles) return;
  const int tw = b % a.tilesW;
  b /= a.tilesW;
  const int th = b % a.tilesH, td = b / a.tilesH;
  const int pz0 = td * a.TZ, py0 = th * a.TY, px0 = tw * a.TXT * 16;
  const int iz0 = pz0 * a.sz - a.pz, iy0 = py0 * a.sy - a.py, ix0 = px0 * a.sx - a.px;

  int base[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int tau = wave * PT + pt;
    const int xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
    base[pt] = (((zt * a.sz) * a.TYI + yt * a.sy) * a.TXI + (xt * 16 + j) * a.sx) * CIS + (4 * g) % CI;
  }
  const int sub = (4 * g) / CI;

  floatx4 acc[CT][PT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int NP = a.TZI * a.TYI * a.TXI, NU = cls.NU;
  // tap table of this class -> LDS (pre-multiplied by the position stride) so the K loop has no dependent global load
  float4 *wl = lds4 + ((size_t)NP * CIS) / 4;                           // [NU][CT][64] packed weights of the current pass
  int *tapl = reinterpret_cast<int *>(wl + (size_t)a.nuMax * CT * 64);  // [NU*TPC] tap offsets (floats)
  for (int i = tid; i < NU * TPC; i += kConvThreads) tapl[i] = a.tapoff[cls.tap_base + i] * CIS;
  const unsigned n_w = (unsigned)NU * CT * 64;
  const int *tp = tapl + sub;
  const unsigned total = (unsigned)NP * C4;
  for (int p = 0; p < a.npass; ++p) {
    // Staging waves get issue priority over co-resident workgroups' K loops: the sooner their loads are in flight the
    // sooner this workgroup can feed the MFMA pipe; the K loop of the neighbour fills the remaining issue slots.
    // (The opposite assignment -- priority to the K loop -- measured 6 % slower.)
    __builtin_amdgcn_s_setprio(2);
    // ---- stage CI channels of the input halo tile into LDS (zero outside the tensor).  Loads are issued in
    // batches of kStageBatch per lane BEFORE the first LDS write so their HBM/L2 latencies overlap. ----
    constexpr int kStageBatch = CT >= 4 ? 6 : 12;  // normally the whole stage: one exposed HBM/L2 latency per pass
#ifdef DR_ABL_NO_STAGE
    if (a.npass < 0)  // ablation build: no staging at all (results are garbage)
#endif
    for (unsigned e0 = 0; e0 < total; e0 += kConvThreads * kStageBatch) {
      float4 v[kStageBatch];
      int dst[kStageBatch];
#pragma unroll
      for (int k = 0; k < kStageBatch; ++k) {
        const unsigned e = e0 + k * kConvThreads + tid;
        const unsigned pos = e / C4, c4 = e - pos * C4;
        // exact for pos < 2^16; a divisor of 1 has no 32-bit magic (2^32), the host stores 0 for it
        const unsigned t = a.magicX ? __umulhi(pos, a.magicX) : pos, x = pos - t * a.TXI;
        const unsigned z = a.magicY ? __umulhi(t, a.magicY) : t, y = t - z * a.TYI;
        const int gz = iz0 + (int)z, gy = iy0 + (int)y, gx = ix0 + (int)x;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        dst[k] = e < total ? (int)(pos * CIS + c4 * 4) : -1;
        if (e < total && gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW)
          v[k] = *reinterpret_cast<const float4 *>(a.in + (((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + p * CI + c4 * 4);
      }
#pragma unroll
      for (int k = 0; k < kStageBatch; ++k)
        if (dst[k] >= 0) *reinterpret_cast<float4 *>(lds + dst[k]) = v[k];
    }
    {  // this pass's packed weights -> LDS, so that the K loop below touches no global memory
      const float4 *wsrc = a.wpk + cls.w_base + ((size_t)p * NU * a.ctTot + ct0) * 64;
      constexpr int kWB = 8;
#ifdef DR_ABL_NO_STAGE
      if (a.npass < 0)
#endif
      for (unsigned e0 = 0; e0 < n_w; e0 += kConvThreads * kWB) {
        float4 v[kWB];
#pragma unroll
        for (int k = 0; k < kWB; ++k) {
          const unsigned e = e0 + k * kConvThreads + tid;  // = (u*CT + ct)*64 + l
          const unsigned u = e / (CT * 64), r = e - u * (CT * 64);
          v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e < n_w) v[k] = wsrc[(size_t)u * a.ctTot * 64 + r];
        }
#pragma unroll
        for (int k = 0; k < kWB; ++k) {
          const unsigned e = e0 + k * kConvThreads + tid;
          if (e < n_w) wl[e] = v[k];
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
#ifndef DR_ABL_NO_KLOOP
    conv_kloop<CT, PT>(lds, wl, tp, TPC, NU, lane, base, acc);
#endif
    __syncthreads();
  }

  conv_epilogue<CT, PT>(a, cls, acc, wave, j, g, ct0, pz0, py0, px0);
}

// ------------------------------------------------------------------------------------------------
// Host side: logical layer description -> packed weights, tap table, tile plan, launches.

enum ConvMode { CONV_NORMAL = 0, CONV_XPAIR = 1, CONV_X8 = 2 };

struct ConvLayer {  // logical description (torch semantics)
  int Cin = 0, Cout = 0;
  int kd = 1, kh = 1, kw = 1;
  int sd = 1, sh = 1, sw = 1;
  bool transposed = false;          // ConvTranspose3d(k=3, pad=1, output_padding = stride-1)
  const float *weight = nullptr;    // (Cout,Cin,kd,kh,kw) or transposed (Cin,Cout,kd,kh,kw)
  std::vector<float> scale, bias;   // per Cout (folded BN / conv bias); empty -> 1 / 0
  bool relu = false;
};

struct ConvLaunch {
  ConvArgs args;
  int ci, ct, pt;
  dim3 grid;
  size_t lds_bytes;
  double flops;  // useful (algorithmic) flops of this launch
};

struct DimTaps {               // per-axis decomposition of one parity class
  std::vector<int> t, off;     // kernel index, input offset (>= 0)
  int s = 1, p = 0, om = 1, oo = 0, npos = 0;
};

// Per-axis tap lists.  Normal conv: in = pos*s - pad + t.  Transposed stride 1: out[o] = sum_t x[o+1-t] w[t].
// Transposed stride 2 (k=3, pad=1, output_padding=1):  out[2m] = x[m] w[1];  out[2m+1] = x[m] w[2] + x[m+1] w[0].
// All 2^d output parities of a position m read the same 2^d input neighbourhood, so the layer runs as ONE stride-1
// convolution with a 2-wide kernel per strided axis and npar*Cout output rows (row = parity*Cout + channel, zero
// weight where a parity does not use an offset); the epilogue scatters row groups to out[2m + parity].
inline int parity_kernel_index(int parity, int off) {  // -1: this (parity, offset) pair carries no weight
  if (parity == 0) return off == 0 ? 1 : -1;
  return off == 0 ? 2 : 0;
}
inline std::vector<DimTaps> axis_classes(int k, int s, bool transposed, int in_size) {
  std::vector<DimTaps> r;
  if (!transposed) {
    DimTaps d;
    for (int t = 0; t < k; ++t) { d.t.push_back(t); d.off.push_back(t); }
    d.s = s; d.p = k / 2; d.npos = (in_size + 2 * (k / 2) - k) / s + 1;
    r.push_back(d);
  } else if (k == 1) {
    DimTaps d; d.t = {0}; d.off = {0}; d.npos = in_size; r.push_back(d);
  } else if (s == 1) {
    DimTaps d;
    for (int t = 0; t < k; ++t) { d.t.push_back(t); d.off.push_back(2 - t); }
    d.p = 1; d.npos = in_size;
    r.push_back(d);
  } else {
    // dense parity form: input offsets {0,1}; which kernel index a (parity, offset) pair selects is resolved when the
    // weights are packed (parity_kernel_index), the two parities of this axis become separate output ROWS
    DimTaps d; d.t = {0, 1}; d.off = {0, 1}; d.om = 2; d.oo = 0; d.npos = in_size; r.push_back(d);
  }
  return r;
}

struct ConvPlanOut {
  std::vector<ConvLaunch> launches;
  int outD, outH, outW;
  int ncand = 0;  // feasible (CI, PT, CT, tile) candidates the planner ranked
};

struct DeviceArena {  // owns small device buffers created while planning (weights, tables)
  std::vector<void *> ptrs;
  template <class T>
  T *upload(const std::vector<T> &h) {
    T *d = dalloc<T>(h.size());
    DR_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    ptrs.push_back(d);
    return d;
  }
  ~DeviceArena() { for (void *p : ptrs) (void)hipFree(p); }
};

// Measured-best plans for known layer shapes (tools/tune_conv.sh -> conv_tuned.h); anything else uses the cost model.
struct ConvTuned {
  int Cin, Cout, kd, kh, kw, sd, sh, sw, transposed, mode, inD, inH, inW;  // layer signature
  int ci, ct, pt, tz, ty, txt;                                              // plan: channel pass, row tiles, position tiles per wave, tile shape (txt in 16s)
};
#include "conv_tuned.h"

inline bool conv_instance_exists(int ci, int ct) {
  return (ci == 4 && ct == 1) || ((ci == 8 || ci == 16) && (ct == 1 || ct == 2 || ct == 4));
}

inline ConvPlanOut plan_conv(const ConvLayer &L, ConvMode mode, const float *in, int inD, int inH, int inW, int inC,
                             float *out, const float *add, int add_mode, DeviceArena &arena, int rank = 0) {
  // rank: which candidate of the cost model's ranking to build (0 = its choice); used by the engine's autotuner
  if (L.Cin % 4 != 0 || inC < L.Cin) fail(DR_ERR_ARG, "plan_conv: Cin=%d must be a multiple of 4 (tensor C=%d)", L.Cin, inC);
  auto cz = axis_classes(L.kd, L.sd, L.transposed, inD);
  auto cy = axis_classes(L.kh, L.sh, L.transposed, inH);
  auto cx = axis_classes(L.kw, L.sw, L.transposed, inW);
  ConvPlanOut R;
  R.outD = L.transposed ? inD * L.sd : cz[0].npos;
  R.outH = L.transposed ? inH * L.sh : cy[0].npos;
  R.outW = L.transposed ? inW * L.sw : cx[0].npos;
  int rows, rows_valid, outWv = R.outW, outCv;
  if (mode == CONV_XPAIR) {
    if (L.transposed || L.sw != 1 || L.Cout != 8 || (R.outW & 1)) fail(DR_ERR_ARG, "XPAIR needs Cout=8, stride 1, even W");
    rows = 16; rows_valid = 16; outWv = R.outW / 2; outCv = 16;
  } else if (mode == CONV_X8) {
    if (L.transposed || L.sw != 1 || L.Cout != 1 || (R.outW & 7)) fail(DR_ERR_ARG, "X8 needs Cout=1, stride 1, W%%8==0");
    rows = 16; rows_valid = 8; outWv = R.outW / 8; outCv = 8;
  } else {
    rows = cdiv(L.Cout, 16) * 16; rows_valid = L.Cout; outCv = L.Cout;
    if (L.Cout % 4) fail(DR_ERR_ARG, "plan_conv: Cout=%d must be a multiple of 4", L.Cout);
  }
  // transposed: parity bits of the strided axes, enumerated (z, y, x) -> par_map holds 3 bits (z<<2|y<<1|x) per parity
  int npar = 1, par_map = 0;
  const bool strided[3] = {L.transposed && L.sd == 2, L.transposed && L.sh == 2, L.transposed && L.sw == 2};
  if (L.transposed) {
    if (mode != CONV_NORMAL) fail(DR_ERR_ARG, "plan_conv: transposed layers use CONV_NORMAL");
    for (int d = 0; d < 3; ++d) if (strided[d]) npar *= 2;
    for (int q = 0; q < npar; ++q) {
      int bits = 0, rem = q;
      for (int d = 2; d >= 0; --d) if (strided[d]) { bits |= (rem & 1) << (2 - d); rem >>= 1; }
      par_map |= bits << (3 * q);
    }
    rows_valid = npar * L.Cout; rows = cdiv(rows_valid, 16) * 16;
  }
  const int CTtot = rows / 16;
  const int shifts = mode == CONV_XPAIR ? 2 : (mode == CONV_X8 ? 8 : 1);
  if (shifts > 1) {  // widen the x taps: in = pos*shifts - pad + t', t' in [0, kw + shifts - 1)
    DimTaps &X = cx[0];
    X.t.clear(); X.off.clear();
    for (int t = 0; t < L.kw + shifts - 1; ++t) { X.t.push_back(t); X.off.push_back(t); }
    X.s = shifts; X.npos = outWv;
  }
  // geometry shared by all parity classes: strides, padding, extents = union over classes
  const int SZ = cz[0].s, SY = cy[0].s, SX = cx[0].s, PZ = cz[0].p, PY = cy[0].p, PX = cx[0].p;
  const int nPD = cz[0].npos, nPH = cy[0].npos, nPW = cx[0].npos;
  int exz = 0, exy = 0, exx = 0;
  for (auto &c : cz) for (int o : c.off) exz = std::max(exz, o + 1);
  for (auto &c : cy) for (int o : c.off) exy = std::max(exy, o + 1);
  for (auto &c : cx) for (int o : c.off) exx = std::max(exx, o + 1);
  struct Cls { const DimTaps *z, *y, *x; int ntaps; };
  std::vector<Cls> classes;
  for (auto &Z : cz) for (auto &Y : cy) for (auto &X : cx) classes.push_back({&Z, &Y, &X, (int)(Z.t.size() * Y.t.size() * X.t.size())});
  const int ncls = (int)classes.size();

  // ---- plan: channel pass width CI, position tiles per wave PT, tile shape, output-row split ----
  static const int cand16[][3] = {{1, 1, 16}, {1, 2, 8}, {1, 4, 4}, {1, 8, 2}, {1, 16, 1}, {2, 1, 8}, {2, 2, 4},
                                  {2, 4, 2}, {2, 8, 1}, {4, 1, 4}, {4, 2, 2}, {4, 4, 1}, {8, 1, 2}, {8, 2, 1}, {16, 1, 1}};
  static const int cand4[][3] = {{1, 1, 4}, {1, 2, 2}, {1, 4, 1}, {2, 1, 2}, {2, 2, 1}, {4, 1, 1}};
  [[maybe_unused]] double best = 1e300;
  int CI = 0, PT = 0, CT = 0, TZ = 0, TY = 0, TXT = 0, TZI = 0, TYI = 0, TXI = 0;
  struct Cand { double cost; int ci, pt, ct, tz, ty, txt, tzi, tyi, txi; };
  std::vector<Cand> cands;
  for (int ci : {16, 8, 4}) {
    if (L.Cin % ci || (ci == 4 && L.Cin != 4)) continue;
    const int npass = L.Cin / ci, tpc = 16 / ci;
    double chunks = 0;  // K chunks per pass summed over classes
    for (auto &c : classes) chunks += cdiv(c.ntaps, tpc);
    for (int pt : {4, 1}) {
      const int (*cand)[3] = pt == 4 ? cand16 : cand4;
      const int ncand = pt == 4 ? 15 : 6;
      for (int k = 0; k < ncand; ++k) {
        const int *c = cand[k];
        const int tzi = (c[0] - 1) * SZ + exz, tyi = (c[1] - 1) * SY + exy, txi = (c[2] * 16 - 1) * SX + exx;
        int nu_max = 0;
        for (auto &cc : classes) nu_max = std::max(nu_max, cdiv(cc.ntaps, tpc));
        const double tiles = (double)cdiv(nPD, c[0]) * cdiv(nPH, c[1]) * cdiv(nPW, c[2] * 16);
        for (int ct : {4, 2, 1}) {
          if (CTtot % ct || !conv_instance_exists(ci, ct)) continue;
          const size_t bytes = (size_t)tzi * tyi * txi * (ci + 4) * 4 + (size_t)nu_max * ct * 1024 + (size_t)nu_max * tpc * 4 + 64;
          if (bytes > kConvMaxLds) continue;
          const int split = CTtot / ct;
          // cost model (cycles): MFMA issue, staging, and a latency floor per chunk; see DESIGN.md
          const double wg_per_cu = std::max(1.0, std::min({(double)(kConvMaxLds / bytes), 8.0, (ct == 4 && pt == 4) ? 5.0 : 8.0}));
          const double stage = npass * (((double)tzi * tyi * txi * (ci / 4) + (chunks / ncls) * ct * 64.0) / 256.0 * 60.0 + 900.0);
          const double chunk_mfma = 4.0 * ct * pt * 32.0;
          const double n_wg = tiles * split;  // per class
          const double mfma_total = n_wg * npass * chunks * chunk_mfma, stage_total = n_wg * ncls * stage;
          const double lat_wg = npass * (chunks / ncls) * std::max(chunk_mfma, 160.0) + stage;
          const double waves = std::ceil(n_wg * ncls / (256.0 * wg_per_cu));
          const double thr = (mfma_total + stage_total) / 256.0 / (wg_per_cu >= 2 ? 0.8 : 0.5);
          const double cost = std::max(thr, waves * lat_wg);
          cands.push_back({cost, ci, pt, ct, c[0], c[1], c[2], tzi, tyi, txi});
        }
      }
    }
  }
  if (!cands.empty()) {
    std::stable_sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.cost < b.cost; });
    if (rank == 0 && !getenv("DR_CONV_NO_TUNED")) {  // a measured plan for exactly this layer moves to the front
      for (const ConvTuned &t : kConvTuned) {
        if (t.Cin != L.Cin || t.Cout != L.Cout || t.kd != L.kd || t.kh != L.kh || t.kw != L.kw || t.sd != L.sd || t.sh != L.sh || t.sw != L.sw ||
            t.transposed != (L.transposed ? 1 : 0) || t.mode != (int)mode || t.inD != inD || t.inH != inH || t.inW != inW) continue;
        for (size_t i = 0; i < cands.size(); ++i) {
          const Cand &k = cands[i];
          if (k.ci == t.ci && k.ct == t.ct && k.pt == t.pt && k.tz == t.tz && k.ty == t.ty && k.txt == t.txt) { std::swap(cands[0], cands[i]); break; }
        }
        break;
      }
    }
    const Cand &k = cands[std::min<size_t>(rank < 0 ? 0 : rank, cands.size() - 1)];
    best = k.cost; CI = k.ci; PT = k.pt; CT = k.ct; TZ = k.tz; TY = k.ty; TXT = k.txt; TZI = k.tzi; TYI = k.tyi; TXI = k.txi;
    R.ncand = (int)cands.size();
  }
  if (!CI) fail(DR_ERR_ARG, "plan_conv: no kernel instance / tile shape for Cin=%d Cout=%d", L.Cin, L.Cout);
  const int npass = L.Cin / CI, TPC = 16 / CI, CIS = CI + 4;

  // per-row epilogue affine
  std::vector<float> sc(rows, 1.f), bi(rows, 0.f);
  for (int r = 0; r < rows_valid; ++r) {
    const int c = L.transposed ? r % L.Cout : (mode == CONV_NORMAL ? r : (mode == CONV_XPAIR ? (r & 7) : 0));
    if (!L.scale.empty()) sc[r] = L.scale[c];
    if (!L.bias.empty()) bi[r] = L.bias[c];
  }
  auto weight_at = [&](int co, int ci, int tz, int ty, int tx) -> float {
    if (!L.transposed) return L.weight[((((size_t)co * L.Cin + ci) * L.kd + tz) * L.kh + ty) * L.kw + tx];
    return L.weight[((((size_t)ci * L.Cout + co) * L.kd + tz) * L.kh + ty) * L.kw + tx];
  };

  // ---- tap tables (LDS position offsets) and packed weights, class after class ----
  std::vector<int> tapoff;
  std::vector<float> pk;
  std::vector<ConvClass> cls(ncls);
  double flops = 0;
  for (int ic = 0; ic < ncls; ++ic) {
    const DimTaps &Z = *classes[ic].z, &Y = *classes[ic].y, &X = *classes[ic].x;
    const int ntz = (int)Z.t.size(), nty = (int)Y.t.size(), ntx = (int)X.t.size(), ntaps = classes[ic].ntaps;
    const int NU = cdiv(ntaps, TPC);
    cls[ic].NU = NU; cls[ic].tap_base = (int)tapoff.size(); cls[ic].w_base = (int)(pk.size() / 4);
    cls[ic].ooz = Z.oo; cls[ic].ooy = Y.oo; cls[ic].oox = X.oo;
    std::vector<int> tz(ntaps), ty(ntaps), tx(ntaps);
    const size_t t0 = tapoff.size();
    tapoff.resize(t0 + (size_t)NU * TPC, 0);
    {
      int n = 0;
      for (int iz = 0; iz < ntz; ++iz) for (int iy = 0; iy < nty; ++iy) for (int ix = 0; ix < ntx; ++ix, ++n) {
        tapoff[t0 + n] = (Z.off[iz] * TYI + Y.off[iy]) * TXI + X.off[ix];
        tz[n] = Z.t[iz]; ty[n] = Y.t[iy]; tx[n] = X.t[ix];
      }
    }
    const size_t w0 = pk.size();
    pk.resize(w0 + (size_t)npass * NU * CTtot * 64 * 4, 0.f);
    for (int p = 0; p < npass; ++p) for (int u = 0; u < NU; ++u) for (int ct = 0; ct < CTtot; ++ct)
      for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) {
        const int g = l >> 4, i = l & 15, k16 = 4 * g + s;
        const int tap = u * TPC + k16 / CI, cin = p * CI + k16 % CI, row = ct * 16 + i;
        float v = 0.f;
        if (tap < ntaps && row < rows_valid) {
          if (L.transposed) {
            const int q = row / L.Cout, co = row % L.Cout, bits = (par_map >> (3 * q)) & 7;
            const int offs[3] = {tz[tap], ty[tap], tx[tap]};  // for transposed layers DimTaps::t carries the input offset
            int kk[3];
            bool ok = true;
            for (int d = 0; d < 3; ++d) {
              if (strided[d]) kk[d] = parity_kernel_index((bits >> (2 - d)) & 1, offs[d]);
              else kk[d] = offs[d];  // stride-1 axis: DimTaps::t is the kernel index already (k == 1 or the 3-tap flip)
              ok = ok && kk[d] >= 0;
            }
            if (ok) v = weight_at(co, cin, kk[0], kk[1], kk[2]);
          } else if (mode == CONV_NORMAL) v = weight_at(row, cin, tz[tap], ty[tap], tx[tap]);
          else {
            const int shift = mode == CONV_XPAIR ? (row >> 3) : row, co = mode == CONV_XPAIR ? (row & 7) : 0;
            const int kx = tx[tap] - shift;
            if (kx >= 0 && kx < L.kw) v = weight_at(co, cin, tz[tap], ty[tap], kx);
          }
        }
        pk[w0 + ((((size_t)p * NU + u) * CTtot + ct) * 64 + l) * 4 + s] = v;
      }
    if (L.transposed) flops += 2.0 * nPD * nPH * nPW * (double)L.kd * L.kh * L.kw * L.Cin * L.Cout;
    else flops += 2.0 * nPD * nPH * nPW * (mode == CONV_NORMAL ? 1 : shifts) * (double)ntz * nty * (mode == CONV_NORMAL ? ntx : L.kw) * L.Cin * L.Cout;
  }

  ConvLaunch cl{};
  ConvArgs &a = cl.args;
  a.in = in; a.out = out; a.wpk = reinterpret_cast<const float4 *>(arena.upload(pk));
  a.scale = arena.upload(sc); a.bias = arena.upload(bi); a.add = add; a.tapoff = arena.upload(tapoff);
  a.cls = arena.upload(cls);
  a.inD = inD; a.inH = inH; a.inW = inW; a.inC = inC;
  a.outD = R.outD; a.outH = R.outH; a.outW = outWv; a.outC = outCv;
  a.nPD = nPD; a.nPH = nPH; a.nPW = nPW;
  a.sz = SZ; a.sy = SY; a.sx = SX; a.pz = PZ; a.py = PY; a.px = PX;
  a.omz = cz[0].om; a.omy = cy[0].om; a.omx = cx[0].om;
  a.par_rows = L.transposed ? L.Cout : 0; a.par_map = par_map;
  a.TZ = TZ; a.TY = TY; a.TXT = TXT; a.TZI = TZI; a.TYI = TYI; a.TXI = TXI;
  a.magicX = TXI == 1 ? 0u : (unsigned)((0x100000000ull + TXI - 1) / TXI);
  a.magicY = TYI == 1 ? 0u : (unsigned)((0x100000000ull + TYI - 1) / TYI);
  if ((size_t)TZI * TYI * TXI >= 65536) fail(DR_ERR_ARG, "plan_conv: halo tile too large");
  a.npass = npass; a.ctTot = CTtot; a.rows_valid = rows_valid; a.relu = L.relu ? 1 : 0;
  a.add_mode = add ? add_mode : 0;
  a.addH = R.outH / 2; a.addW = (mode == CONV_NORMAL ? R.outW : outWv) / 2;
  a.tilesD = cdiv(nPD, TZ); a.tilesH = cdiv(nPH, TY); a.tilesW = cdiv(nPW, TXT * 16);
  cl.ci = CI; cl.ct = CT; cl.pt = PT;
  cl.grid = dim3(8 * cdiv(a.tilesD * a.tilesH * a.tilesW, 8), ncls, CTtot / CT);
  int nu_max = 0;
  for (auto &c : cls) nu_max = std::max(nu_max, c.NU);
  a.nuMax = nu_max;
  cl.lds_bytes = (size_t)TZI * TYI * TXI * CIS * 4 + (size_t)nu_max * CT * 1024 + (size_t)nu_max * TPC * 4 + 64;
  cl.flops = flops;
  R.launches.push_back(cl);
  return R;
}

// MaxDynamicSharedMemorySize is a per-device attribute of each kernel: the opt-in is tracked per (device, instance).
inline void conv_allow_big_lds(const void *fn, std::atomic<unsigned long long> &done, size_t lds_bytes) {
  if (lds_bytes <= 64 * 1024) return;
  int dev = 0;
  DR_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  DR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kConvMaxLds));
  done.fetch_or(bit, std::memory_order_release);
}
template <int CI, int CT, int PT>
inline void launch_conv_inst(const ConvLaunch &c, hipStream_t st) {
  static std::atomic<unsigned long long> done{0};  // bit d: set on device d
  conv_allow_big_lds(reinterpret_cast<const void *>(&k_conv<CI, CT, PT>), done, c.lds_bytes);
  hipLaunchKernelGGL((k_conv<CI, CT, PT>), c.grid, dim3(kConvThreads), c.lds_bytes, st, c.args);
}
inline void launch_conv(const ConvLaunch &c, hipStream_t st) {
#define DR_CONV_CASE(CI_, CT_)                                                  \
  if (c.ci == CI_ && c.ct == CT_) {                                             \
    if (c.pt == 4) launch_conv_inst<CI_, CT_, 4>(c, st);                        \
    else launch_conv_inst<CI_, CT_, 1>(c, st);                                  \
    return;                                                                     \
  }
  DR_CONV_CASE(4, 1)
  DR_CONV_CASE(8, 1)
  DR_CONV_CASE(8, 2)
  DR_CONV_CASE(8, 4)
  DR_CONV_CASE(16, 1)
  DR_CONV_CASE(16, 2)
  DR_CONV_CASE(16, 4)
#undef DR_CONV_CASE
  fail(DR_ERR_ARG, "launch_conv: no instance CI=%d CT=%d", c.ci, c.ct);
}

}  // namespace dr
