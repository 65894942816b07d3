// conv_wino.h -- k_conv_w: the stride-1 3x3 / 3x3x3 layers with the 3-tap kernel axis y evaluated in Winograd's minimal-filtering form
// F(2,3): two output rows from four input rows with 4 products per (x tap, z tap, channel) instead of 6 -- the fp32-legal cut of the
// MFMA work of these layers by a third (VERDICT r3 "no fp32-legal reduction of the convolution arithmetic").
//
//   input transform   (per lane, registers):  v0 = d0 - d2,  v1 = d1 + d2,  v2 = d2 - d1,  v3 = d1 - d3     (d_q = input row 2Y - 1 + q)
//   weights           (host, plan_conv)    :  u0 = g0,  u1 = (g0 + g1 + g2) / 2,  u2 = (g0 - g1 + g2) / 2,  u3 = g2
//   products          (MFMA)               :  m_p += U_p . v_p   over (x tap, z tap, channel): four independent implicit GEMMs
//   output transform  (per lane, registers):  out[2Y] = m0 + m1 + m2,  out[2Y + 1] = m1 - m2 - m3
//
// Everything else is k_conv's (conv_mfma.h): the halo tile staged through registers, this pass's packed weights by LDS-DMA, one wave = PT
// position tiles, operand mapping A = weights (16 rows x 4 K), B = inputs (4 K x 16 consecutive x), the tap table (which now lists the
// (z, x) taps only: the kernel walks the four rows itself), XPAIR for Cout = 8 (rows = 8 channels x 2 adjacent x, 4-wide x window),
// epilogue (folded BN, ReLU, residual / upsample add).  A position tile is 16 x by ONE ROW PAIR; to the planner the layer is a
// stride-(1,2,1) convolution with a 4-row kernel and two output rows per position (a.sy = 2, a.omy = 2).
// The transform is exact in the weights up to one rounding per u_p (computed in double) and costs one rounding per v_p and two per
// output: measured against torch fp32 the error stays at the level of fp32 reassociation (tests/test_conv_gpu.py, bounds unchanged).
#pragma once

namespace dr {

template <int CT, int PT>
__device__ inline void conv_w_load(const float *lds, const float4 *wp, int toff, int r, const int (&base)[PT], int rs, float4 (&av)[4][CT], float4 (&dv)[4][PT]) {
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) av[p][ct] = wp[((r * 4 + p) * CT + ct) * 64];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const float *b = lds + base[pt] + toff;
#pragma unroll
    for (int q = 0; q < 4; ++q) dv[q][pt] = *reinterpret_cast<const float4 *>(b + q * rs);
  }
}
template <int CT, int PT>
__device__ inline void conv_w_anchor(const float4 (&av)[4][CT], const float4 (&dv)[4][PT], int toff) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) asm volatile("" ::"v"(av[p][ct].x), "v"(av[p][ct].y), "v"(av[p][ct].z), "v"(av[p][ct].w));
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) asm volatile("" ::"v"(dv[p][pt].x), "v"(dv[p][pt].y), "v"(dv[p][pt].z), "v"(dv[p][pt].w));
  }
  asm volatile("" ::"v"(toff));
}
// rows d0..d3 -> v0..v3, in place
template <int PT>
__device__ inline void conv_w_transform(float4 (&d)[4][PT]) {
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const float4 d1 = d[1][pt], d2 = d[2][pt];
    d[0][pt] = make_float4(d[0][pt].x - d2.x, d[0][pt].y - d2.y, d[0][pt].z - d2.z, d[0][pt].w - d2.w);
    d[3][pt] = make_float4(d1.x - d[3][pt].x, d1.y - d[3][pt].y, d1.z - d[3][pt].z, d1.w - d[3][pt].w);
    d[1][pt] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
    d[2][pt] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
  }
}
template <int CT, int PT>
__device__ inline void conv_w_mfma(const float4 (&av)[4][CT], const float4 (&v)[4][PT], floatx4 (&acc)[4][CT][PT]) {
  // consecutive MFMAs go to different accumulators: the same one comes round again after 4 * CT * PT instructions
#define DR_W_STEP(S)                                                                                                        \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) _Pragma("unroll") for (int pt = 0; pt < PT; ++pt) \
      acc[p][ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p][ct].S, v[p][pt].S, acc[p][ct][pt], 0, 0, 0);
  DR_W_STEP(x) DR_W_STEP(y) DR_W_STEP(z) DR_W_STEP(w)
#undef DR_W_STEP
}
// K loop of one channel pass: NR row-group chunks, each = 4 points x 4 MFMAs x CT x PT; the raw rows and the four weight fragments of
// chunk r + 1 are fetched under the MFMAs of chunk r (two register sets, the loop unrolled by two, as conv_kloop).
template <int CT, int PT>
__device__ inline void conv_w_kloop(const float *lds, const float4 *wl, const int *tp, int TPC, int NR, int lane, const int (&base)[PT], int rs,
                                    floatx4 (&acc)[4][CT][PT]) {
  const float4 *wp = wl + lane;
  float4 a0[4][CT], d0[4][PT], a1[4][CT], d1[4][PT];
  int tA = tp[0], tB = tp[min(1, NR - 1) * TPC];
  conv_w_load<CT, PT>(lds, wp, tA, 0, base, rs, a0, d0);
  int r = 0;
  for (; r + 1 < NR; r += 2) {
    conv_w_load<CT, PT>(lds, wp, tB, r + 1, base, rs, a1, d1);
    tA = tp[min(r + 2, NR - 1) * TPC];
    __builtin_amdgcn_sched_barrier(0);
    conv_w_transform<PT>(d0);
    conv_w_mfma<CT, PT>(a0, d0, acc);
    __builtin_amdgcn_sched_barrier(0);
    conv_w_anchor<CT, PT>(a1, d1, tA);
    conv_w_load<CT, PT>(lds, wp, tA, min(r + 2, NR - 1), base, rs, a0, d0);
    tB = tp[min(r + 3, NR - 1) * TPC];
    __builtin_amdgcn_sched_barrier(0);
    conv_w_transform<PT>(d1);
    conv_w_mfma<CT, PT>(a1, d1, acc);
    __builtin_amdgcn_sched_barrier(0);
    conv_w_anchor<CT, PT>(a0, d0, tB);
  }
  if (r < NR) {
    conv_w_transform<PT>(d0);
    conv_w_mfma<CT, PT>(a0, d0, acc);
  }
}

// out[2Y] and out[2Y + 1] of this lane's 4 rows x 1 column, then k_conv's epilogue arithmetic per output row
template <int CT, int PT>
__device__ inline void conv_w_epilogue(const ConvArgs &a, const ConvClass &cls, floatx4 (&acc)[4][CT][PT], const float4 (&scv)[CT], const float4 (&biv)[CT],
                                       int wave, int j, int g, int ct0, int pz0, int py0, int px0) {
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
      const floatx4 m0 = acc[0][ct][pt], m1 = acc[1][ct][pt], m2 = acc[2][ct][pt], m3 = acc[3][ct][pt];
      const floatx4 o[2] = {(m0 + m1) + m2, (m1 - m2) - m3};
      const float4 sc = scv[ct], bi = biv[ct];
      const int oz = qz * a.omz + cls.ooz, ox = qx * a.omx + cls.oox;
      size_t obase[2];  // both rows' residual operands before the first store (`add` may alias `out`)
      float4 ad[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int oy = qy * 2 + r;
        obase[r] = (((size_t)oz * a.outH + oy) * a.outW + ox) * a.outC + c0;
        size_t abase = obase[r];
        if (a.add_mode == 2) abase = (((size_t)oz * a.addH + (oy >> 1)) * a.addW + (ox >> 1)) * a.outC + c0;
        ad[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.add_mode) ad[r] = *reinterpret_cast<const float4 *>(a.add + abase);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float4 v;
        v.x = o[r][0] * sc.x + bi.x;
        v.y = o[r][1] * sc.y + bi.y;
        v.z = o[r][2] * sc.z + bi.z;
        v.w = o[r][3] * sc.w + bi.w;
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.add_mode) { v.x += ad[r].x; v.y += ad[r].y; v.z += ad[r].z; v.w += ad[r].w; }
        *reinterpret_cast<float4 *>(a.out + obase[r]) = v;
      }
    }
  }
}

// grid = (tiles, 1, output-row groups), 4 waves, one tile per workgroup (k_conv's launch form)
template <int CI, int CT, int PT>
// waves per SIMD the register budget is cut for: 4 x 128, 3 x 168, 2 x 256 registers
__global__ __launch_bounds__(kConvThreads, (CT * PT == 1 ? 4 : (CT * PT == 2 ? 3 : 2))) void k_conv_w(const ConvArgs a) {
  extern __shared__ float4 lds4[];
  float *lds = reinterpret_cast<float *>(lds4);
  constexpr int CIS = CI + 4, TPC = 16 / CI, C4 = CI / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const ConvClass cls = a.cls[0];
  const int ct0 = blockIdx.z * CT;
  const int ntiles = a.tilesD * a.tilesH * a.tilesW, per_xcd = (ntiles + 7) >> 3;  // XCD k takes the k-th contiguous range of tiles (k_conv)
  int b = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (b >= ntiles) return;
  const int tw = b % a.tilesW;
  b /= a.tilesW;
  const int th = b % a.tilesH, td = b / a.tilesH;
  const int pz0 = td * a.TZ, py0 = th * a.TY, px0 = tw * a.TXT * 16;
  const int iz0 = pz0 * a.sz - a.pz, iy0 = py0 * a.sy - a.py, ix0 = px0 * a.sx - a.px;
  const int rs = a.TXI * CIS;  // LDS floats between two rows of the staged tile

  int base[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int tau = wave * PT + pt;
    const int xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
    base[pt] = (((zt * a.sz) * a.TYI + yt * a.sy) * a.TXI + (xt * 16 + j) * a.sx) * CIS + (4 * g) % CI;
  }
  const int sub = (4 * g) / CI;

  floatx4 acc[4][CT][PT];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[p][ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int NP = a.TZI * a.TYI * a.TXI, NU = cls.NU, NR = NU >> 2;  // NU counts weight chunks: four (one per point) per row-group chunk
  float4 *wl = lds4 + ((size_t)NP * CIS) / 4;                           // [NR][4][CT][64] packed weights of the current pass
  int *tapl = reinterpret_cast<int *>(wl + (size_t)a.nuMax * CT * 64);  // [NR * TPC] tap offsets (floats)
  for (int i = tid; i < NR * TPC; i += kConvThreads) tapl[i] = a.tapoff[cls.tap_base + i] * CIS;
  const int *tp = tapl + sub;
  const unsigned total = (unsigned)NP * C4;
  for (int p = 0; p < a.npass; ++p) {
    __builtin_amdgcn_s_setprio(2);
    const float4 *wsrc = a.wpk + cls.w_base + ((size_t)p * NU * a.ctTot + ct0) * 64;
    for (int e = wave; e < NU * CT; e += kConvThreads / 64) {  // piece e = u * CT + ct: 64 lanes x 16 B, contiguous on both sides
      const int u = e / CT, ct = e - u * CT;
      conv_a_dma16(wsrc + ((size_t)u * a.ctTot + ct) * 64 + lane, __builtin_amdgcn_readfirstlane(conv_a_lds_addr(wl + (size_t)e * 64)));
    }
    constexpr int kStageBatch = 12;
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
        dst[k] = e < total ? (int)(pos * CIS + c4 * 4) : -1;
        if (e < total && gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW)
          v[k] = *reinterpret_cast<const float4 *>(a.in + (((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + (size_t)p * a.pass_stride + c4 * 4);
      }
#pragma unroll
      for (int k = 0; k < kStageBatch; ++k)
        if (dst[k] >= 0) *reinterpret_cast<float4 *>(lds + dst[k]) = v[k];
    }
    conv_a_wait_dma();
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    conv_w_kloop<CT, PT>(lds, wl, tp, TPC, NR, lane, base, rs, acc);
    __syncthreads();
  }
  float4 scv[CT], biv[CT];
  conv_load_affine<CT>(a, g, ct0, scv, biv);
  conv_w_epilogue<CT, PT>(a, cls, acc, scv, biv, wave, j, g, ct0, pz0, py0, px0);
}

}  // namespace dr
