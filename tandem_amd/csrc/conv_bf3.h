// conv_bf3.h -- k_conv_b: the convolution kernel's opt-in reduced-precision form (DR_CONV_BF16X3=1), included by conv_mfma.h.
//
// fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate on gfx950, and the depth pipeline's parity bounds are set by
// fp32 reassociation, not by the last bits of every product.  Here both operands of every product are split into two bf16 terms,
//     x = x_h + x_l,   x_h = bf16(x),   x_l = bf16(x - x_h)        (round to nearest even; |x - x_h - x_l| <= 2^-17 |x|)
// and the three leading products  w_h x_h + w_l x_h + w_h x_l  are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (products of
// bf16 values are exact in fp32; the dropped w_l x_l term is <= 2^-16 relative).  tools/study_split_bf16.py evaluates the whole depth
// pipeline under this arithmetic on the CPU: the depth maps stay inside the bounds the fp32 path is held to (mean 2e-5 m, max 2e-4 m
// against 1e-4 / 5e-2), while plain bf16 or two terms do not (profiles/r03_split_bf16_study.txt).
//
// The kernel is k_conv (conv_mfma.h) with two changes; planner, tap tables, parity classes, tile geometry and epilogue are shared:
//   * staging splits: the record of a staged position keeps its size (CI + 4 floats) and holds [CI hi bf16 | CI lo bf16 | pad], so
//     LDS geometry and tap offsets are the fp32 kernel's;
//   * the K loop walks 32-wide chunks: lane (j = l & 15, g = l >> 4) owns K = 32 u + 8 g .. + 7 = eight consecutive channels of one
//     tap, i.e. ONE ds_read_b128 for the hi and one for the lo fragment, and the packed weights (split on the host, plan_conv) come
//     as a hi and a lo fragment per (chunk, row tile): three MFMAs of 16 cycles replace eight of 32.
// HBM formats do not change: activations are fp32 tensors, every other kernel is untouched.
//
// Status: written and checked against a host emulation of its data flow (tests/cpp/conv_emul.hip) at the end of round 3, when the
// round's GPU time was spent; NOT yet run on a GPU.  It is reachable only through DR_CONV_BF16X3=1.
#pragma once
// (included inside namespace dr, after k_conv and the LDS-DMA helpers)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// (a, b) -> packed bf16 pairs: hi = (bf16(a), bf16(b)), lo = the bf16 of what the hi terms leave; element 0 in the low half
__device__ inline void bf3_split2(float a, float b, unsigned &hi, unsigned &lo) {
  const bf16x2 h = __builtin_convertvector(floatx2{a, b}, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float ha = __uint_as_float(hi << 16), hb = __uint_as_float(hi & 0xffff0000u);
  const bf16x2 l = __builtin_convertvector(floatx2{a - ha, b - hb}, bf16x2);
  lo = __builtin_bit_cast(unsigned, l);
}

template <int CT, int PT>
__device__ inline void conv_b_load(const char *ldsb, const float4 *wp, int toffb, int u, const int (&baseb)[PT], int lo_off, float4 (&ah)[CT],
                                   float4 (&al)[CT], float4 (&bh)[PT], float4 (&bl)[PT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    ah[ct] = wp[((u * CT + ct) * 2 + 0) * 64];
    al[ct] = wp[((u * CT + ct) * 2 + 1) * 64];
  }
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    bh[pt] = *reinterpret_cast<const float4 *>(ldsb + baseb[pt] + toffb);
    bl[pt] = *reinterpret_cast<const float4 *>(ldsb + baseb[pt] + toffb + lo_off);
  }
}
template <int CT, int PT>
__device__ inline void conv_b_mfma(const float4 (&ah)[CT], const float4 (&al)[CT], const float4 (&bh)[PT], const float4 (&bl)[PT], floatx4 (&acc)[CT][PT]) {
  // the two small terms first; consecutive MFMAs go to different accumulators
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
      acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[ct]), __builtin_bit_cast(bf16x8, bh[pt]), acc[ct][pt], 0, 0, 0);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
      acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[ct]), __builtin_bit_cast(bf16x8, bl[pt]), acc[ct][pt], 0, 0, 0);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
      acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[ct]), __builtin_bit_cast(bf16x8, bh[pt]), acc[ct][pt], 0, 0, 0);
}
template <int CT, int PT>
__device__ inline void conv_b_anchor(const float4 (&ah)[CT], const float4 (&al)[CT], const float4 (&bh)[PT], const float4 (&bl)[PT], int toff) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    asm volatile("" ::"v"(ah[ct].x), "v"(ah[ct].y), "v"(ah[ct].z), "v"(ah[ct].w));
    asm volatile("" ::"v"(al[ct].x), "v"(al[ct].y), "v"(al[ct].z), "v"(al[ct].w));
  }
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    asm volatile("" ::"v"(bh[pt].x), "v"(bh[pt].y), "v"(bh[pt].z), "v"(bh[pt].w));
    asm volatile("" ::"v"(bl[pt].x), "v"(bl[pt].y), "v"(bl[pt].z), "v"(bl[pt].w));
  }
  asm volatile("" ::"v"(toff));
}
// conv_kloop's schedule (operands of chunk u + 1 fetched under the MFMAs of chunk u, two register sets used alternately)
template <int CT, int PT>
__device__ inline void conv_b_kloop(const char *ldsb, const float4 *wl, const int *tp, int TPC, int NU, int lane, const int (&baseb)[PT], int lo_off,
                                    floatx4 (&acc)[CT][PT]) {
  const float4 *wp = wl + lane;
  float4 ah0[CT], al0[CT], bh0[PT], bl0[PT], ah1[CT], al1[CT], bh1[PT], bl1[PT];
  int tA = tp[0], tB = tp[min(1, NU - 1) * TPC];
  conv_b_load<CT, PT>(ldsb, wp, tA, 0, baseb, lo_off, ah0, al0, bh0, bl0);
  int u = 0;
  for (; u + 1 < NU; u += 2) {
    conv_b_load<CT, PT>(ldsb, wp, tB, u + 1, baseb, lo_off, ah1, al1, bh1, bl1);
    tA = tp[min(u + 2, NU - 1) * TPC];
    __builtin_amdgcn_sched_barrier(0);
    conv_b_mfma<CT, PT>(ah0, al0, bh0, bl0, acc);
    __builtin_amdgcn_sched_barrier(0);
    conv_b_anchor<CT, PT>(ah1, al1, bh1, bl1, tA);
    conv_b_load<CT, PT>(ldsb, wp, tA, min(u + 2, NU - 1), baseb, lo_off, ah0, al0, bh0, bl0);
    tB = tp[min(u + 3, NU - 1) * TPC];
    __builtin_amdgcn_sched_barrier(0);
    conv_b_mfma<CT, PT>(ah1, al1, bh1, bl1, acc);
    __builtin_amdgcn_sched_barrier(0);
    conv_b_anchor<CT, PT>(ah0, al0, bh0, bl0, tB);
  }
  if (u < NU) conv_b_mfma<CT, PT>(ah0, al0, bh0, bl0, acc);  // odd chunk count: set 0 holds the last chunk
}

// grid = (tiles, parity classes, output-row groups), 4 waves, one tile per workgroup: k_conv's launch form.
template <int CI, int CT, int PT>
__global__ __launch_bounds__(kConvThreads) void k_conv_b(const ConvArgs a) {
  static_assert(CI == 8 || CI == 16, "a lane's K group is eight consecutive channels of one tap");
  extern __shared__ float4 lds4[];
  char *ldsb = reinterpret_cast<char *>(lds4);
  constexpr int CIS = CI + 4;      // LDS floats per staged position: [CI hi bf16 | CI lo bf16 | 16 bytes of padding]
  constexpr int RB = CIS * 4;      // ... in bytes
  constexpr int TPC = 32 / CI;     // taps per 32-wide K chunk
  constexpr int C4 = CI / 4;
  constexpr int LO = 2 * CI;       // byte offset of the lo half inside a record
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const ConvClass cls = a.cls[blockIdx.y];
  const int ct0 = blockIdx.z * CT;

  const int ntiles = a.tilesD * a.tilesH * a.tilesW, per_xcd = (ntiles + 7) >> 3;  // XCD-aware tile order, as k_conv
  int b = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (b >= ntiles) return;
  const int tw = b % a.tilesW;
  b /= a.tilesW;
  const int th = b % a.tilesH, td = b / a.tilesH;
  const int pz0 = td * a.TZ, py0 = th * a.TY, px0 = tw * a.TXT * 16;
  const int iz0 = pz0 * a.sz - a.pz, iy0 = py0 * a.sy - a.py, ix0 = px0 * a.sx - a.px;

  int baseb[PT];  // byte address of this lane's K group at tap offset 0
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int tau = wave * PT + pt;
    const int xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
    baseb[pt] = (((zt * a.sz) * a.TYI + yt * a.sy) * a.TXI + (xt * 16 + j) * a.sx) * RB + ((8 * g) % CI) * 2;
  }
  const int sub = (8 * g) / CI;

  floatx4 acc[CT][PT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int NP = a.TZI * a.TYI * a.TXI, NU = cls.NU;
  float4 *wl = lds4 + ((size_t)NP * CIS) / 4;                               // [NU][CT][hi | lo][64] packed weight fragments of the current pass
  int *tapl = reinterpret_cast<int *>(wl + (size_t)a.nuMax * CT * 2 * 64);  // [NU * TPC] tap offsets in bytes
  for (int i = tid; i < NU * TPC; i += kConvThreads) tapl[i] = a.tapoff[cls.tap_base + i] * RB;
  const int *tp = tapl + sub;
  const unsigned total = (unsigned)NP * C4;
  for (int p = 0; p < a.npass; ++p) {
    __builtin_amdgcn_s_setprio(2);
    // this pass's weight fragments by LDS-DMA, requested before the tile is staged (piece e = (u * CT + ct) * 2 + half)
    const float4 *wsrc = a.wpk + cls.w_base + ((size_t)p * NU * a.ctTot + ct0) * 2 * 64;
    for (int e = wave; e < NU * CT * 2; e += kConvThreads / 64) {
      const int h = e & 1, uc = e >> 1, u = uc / CT, ct = uc - u * CT;
      conv_a_dma16(wsrc + (((size_t)u * a.ctTot + ct) * 2 + h) * 64 + lane, __builtin_amdgcn_readfirstlane(conv_a_lds_addr(wl + (size_t)e * 64)));
    }
    // stage CI channels of the halo tile, split on the way: four channels = 8 bytes of hi and 8 bytes of lo
    constexpr int kStageBatch = CT >= 4 ? 6 : 12;
    for (unsigned e0 = 0; e0 < total; e0 += kConvThreads * kStageBatch) {
      float4 v[kStageBatch];
      int dst[kStageBatch];
#pragma unroll
      for (int k = 0; k < kStageBatch; ++k) {
        const unsigned e = e0 + k * kConvThreads + tid;
        const unsigned pos = e / C4, c4 = e - pos * C4;
        const unsigned t = a.magicX ? __umulhi(pos, a.magicX) : pos, x = pos - t * a.TXI;
        const unsigned z = a.magicY ? __umulhi(t, a.magicY) : t, y = t - z * a.TYI;
        const int gz = iz0 + (int)z, gy = iy0 + (int)y, gx = ix0 + (int)x;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        dst[k] = e < total ? (int)(pos * RB + c4 * 8) : -1;
        if (e < total && gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW)
          v[k] = *reinterpret_cast<const float4 *>(a.in + (((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + p * CI + c4 * 4);
      }
#pragma unroll
      for (int k = 0; k < kStageBatch; ++k)
        if (dst[k] >= 0) {
          uint2 hi, lo;
          bf3_split2(v[k].x, v[k].y, hi.x, lo.x);
          bf3_split2(v[k].z, v[k].w, hi.y, lo.y);
          *reinterpret_cast<uint2 *>(ldsb + dst[k]) = hi;
          *reinterpret_cast<uint2 *>(ldsb + dst[k] + LO) = lo;
        }
    }
    conv_a_wait_dma();
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    conv_b_kloop<CT, PT>(ldsb, wl, tp, TPC, NU, lane, baseb, LO, acc);
    __syncthreads();
  }

  float4 scv[CT], biv[CT];
  conv_load_affine<CT>(a, g, ct0, scv, biv);
  conv_epilogue<CT, PT>(a, cls, acc, scv, biv, wave, j, g, ct0, pz0, py0, px0);
}

