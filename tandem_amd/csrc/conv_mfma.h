// conv_mfma.h -- the convolution kernels of the depth pipeline: a tap-table implicit GEMM on
// the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32, exact fp32 == an fmaf chain), in two launch forms that
// share operand mapping, packed weights, tap tables and epilogue:
//   k_conv<CI,CT,PT,FZ>  4 waves, one tile per workgroup, register-staged halo tile (FZ = 8: the staging step computes
//                        FeatureNet's skip.stage3 + upsample instead of copying a tensor);
//   k_conv_a<CI,CT,PT>   8 waves, persistent over (tile, channel pass) units, halo tiles streamed into a second LDS
//                        buffer by LDS-DMA (global_load_lds_dwordx4) under the K loop of the current unit.
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
  // FZ instances: the input tensor is never materialised -- the staging step computes it from FeatureNet's skip pair,
  // in[v][y][x][c] = (W1 . fz_x[v][y][x][0..FZ) + fz_b)[c] + fz_coarse[v][y/2][x/2][c]   (module.py:517-529)
  const float *fz_x, *fz_w, *fz_b, *fz_coarse;
  // k_conv_a (persistent, LDS-DMA staged) only:
  int pass_stride;      // input elements between the channel slices of two passes: CI (the slices interleave inside a position's record) or, for an
                        // input stored as consecutive (D,H,W,CI) sub-tensors (stage 1's cost volume), the size of one sub-tensor
  const float *zero16;  // 16 zero bytes: the source of every staged element outside the tensor
  int a_slots;          // 16-byte LDS slots per tile buffer (multiple of 512)
  int a_wbufs;          // weight buffers: npass when all passes fit (each fetched once per workgroup), else 2 (one per pass in flight)
  int class_loop;       // k_conv, single-pass transposed layers: > 0 = the number of parity classes ONE workgroup walks on its staged tile (grid.y = 1), see k_conv
};

constexpr int kConvThreads = 256;
constexpr size_t kConvMaxLds = 160 * 1024;  // gfx950: 160 KiB LDS per CU, one workgroup may take all of it

// ---- epilogue: folded BN, ReLU, residual / upsample add, one float4 (4 channels) per lane ----
template <int CT>
__device__ inline void conv_load_affine(const ConvArgs &a, int g, int ct0, float4 (&sc)[CT], float4 (&bi)[CT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {  // scale / bias are padded to whole 16-row tiles on the host
    sc[ct] = *reinterpret_cast<const float4 *>(a.scale + (ct0 + ct) * 16 + 4 * g);
    bi[ct] = *reinterpret_cast<const float4 *>(a.bias + (ct0 + ct) * 16 + 4 * g);
  }
}
// The residual / upsample-add operand is FETCHED FOR A GROUP OF UP TO FOUR (position tile, row tile) entries before the first store of the
// group: `add` and `out` may alias (the folded stage-3 head adds in place), so hipcc keeps every load behind the store before it, and the
// per-entry form paid one exposed memory round trip per entry (load, wait, store, load, wait, store ... in the ISA of every add layer).
// ADD = false: the instantiation for layers without a residual operand -- no load and therefore no s_waitcnt vmcnt in it (k_conv_a issues the next
// unit's DMA before it: a counter wait here would wait for those older pieces as well).
template <int CT, int PT, bool ADD = true>
__device__ inline void conv_epilogue(const ConvArgs &a, const ConvClass &cls, floatx4 (&acc)[CT][PT], const float4 (&scv)[CT], const float4 (&biv)[CT],
                                     int wave, int j, int g, int ct0, int pz0, int py0, int px0) {
#ifdef DR_EPILOGUE_PER_ENTRY  // A/B build: the per-entry form
  constexpr int NE = CT * PT, GB = 1;
#else
  constexpr int NE = CT * PT, GB = NE < 4 ? NE : 4;
#endif
#pragma unroll
  for (int e0 = 0; e0 < NE; e0 += GB) {
    size_t obase[GB];
    float4 r[GB];
    bool ok[GB];
#pragma unroll
    for (int k = 0; k < GB; ++k) {
      const int e = e0 + k, pt = e / CT, ct = e - pt * CT;  // (compile-time after unrolling)
      const int tau = wave * PT + pt;
      const int xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
      const int qz = pz0 + zt, qy = py0 + yt, qx = px0 + xt * 16 + j;
      const int c0 = (ct0 + ct) * 16 + 4 * g;
      ok[k] = e < NE && qz < a.nPD && qy < a.nPH && qx < a.nPW && c0 < a.rows_valid;
      int oz = qz * a.omz + cls.ooz, oy = qy * a.omy + cls.ooy, ox = qx * a.omx + cls.oox, ch = c0;
      if (a.par_rows) {  // transposed layer: this lane's 4 rows are 4 channels of output parity q
        const int q = c0 / a.par_rows, bits = (a.par_map >> (3 * q)) & 7;
        ch = c0 - q * a.par_rows;
        oz += (bits >> 2) & 1; oy += (bits >> 1) & 1; ox += bits & 1;
      }
      obase[k] = (((size_t)oz * a.outH + oy) * a.outW + ox) * a.outC + ch;
      size_t abase = obase[k];
      if (a.add_mode == 2) abase = (((size_t)oz * a.addH + (oy >> 1)) * a.addW + (ox >> 1)) * a.outC + ch;
      r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ADD && a.add_mode && ok[k]) r[k] = *reinterpret_cast<const float4 *>(a.add + abase);
    }
#pragma unroll
    for (int k = 0; k < GB; ++k) {
      const int e = e0 + k, pt = e / CT, ct = e - pt * CT;
      if (e >= NE || !ok[k]) continue;
      const float4 sc = scv[ct], bi = biv[ct];
      float4 v;
      v.x = acc[ct][pt][0] * sc.x + bi.x;
      v.y = acc[ct][pt][1] * sc.y + bi.y;
      v.z = acc[ct][pt][2] * sc.z + bi.z;
      v.w = acc[ct][pt][3] * sc.w + bi.w;
      if (a.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (ADD && a.add_mode) { v.x += r[k].x; v.y += r[k].y; v.z += r[k].z; v.w += r[k].w; }
      *reinterpret_cast<float4 *>(a.out + obase[k]) = v;
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

// The K loop for NARROW waves (CT * PT <= 2: the coarse and strided layers, ~25 launches per forward).  Two things the loop above leaves
// on the table there: (1) a chunk is 4 or 8 MFMAs = 128-256 cycles, about one LDS round trip, so operands fetched ONE chunk ahead arrive
// late -- here they are fetched two chunks ahead (three register sets in rotation, the loop unrolled by three); (2) with a single
// accumulator tile the four MFMAs of a chunk depend on each other (40-cycle dependent latency against a 32-cycle issue interval) --
// here they alternate between two accumulators that are added at the end (k_conv_m does the same, conv_march.h).
// `load(tap offset, chunk, av, bv)` fetches one chunk's operands (k_conv and k_conv_a address their tiles differently).
template <int CT, int PT>
__device__ inline void conv_chunk_mfma2(const float4 (&av)[CT], const float4 (&bv)[PT], floatx4 (&acc)[CT][PT], floatx4 (&acc2)[CT][PT]) {
  if constexpr (CT * PT == 1) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].x, bv[0].x, acc[0][0], 0, 0, 0);
    acc2[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].y, bv[0].y, acc2[0][0], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].z, bv[0].z, acc[0][0], 0, 0, 0);
    acc2[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0].w, bv[0].w, acc2[0][0], 0, 0, 0);
  } else conv_chunk_mfma<CT, PT>(av, bv, acc);
}
template <int CT, int PT, class Load>
__device__ inline void conv_kloop_narrow(const int *tp, int TPC, int NU, floatx4 (&acc)[CT][PT], Load load) {
  float4 a0[CT], b0[PT], a1[CT], b1[PT], a2[CT], b2[PT];
  floatx4 acc2[CT][PT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc2[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
  auto tap = [&](int u) { return tp[min(u, NU - 1) * TPC]; };  // (the tap offset of a chunk is itself an LDS read: fetched a chunk before its operands)
  int t0 = tap(0), t1 = tap(1), t2 = tap(2);
  load(t0, 0, a0, b0);
  load(t1, min(1, NU - 1), a1, b1);
  int u = 0;
  for (; u + 2 < NU; u += 3) {  // invariant: set 0 holds chunk u, set 1 chunk u + 1
    load(t2, u + 2, a2, b2);
    t0 = tap(u + 3);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_mfma2<CT, PT>(a0, b0, acc, acc2);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_anchor<CT, PT>(a1, b1, t0);
    load(t0, min(u + 3, NU - 1), a0, b0);
    t1 = tap(u + 4);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_mfma2<CT, PT>(a1, b1, acc, acc2);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_anchor<CT, PT>(a2, b2, t1);
    load(t1, min(u + 4, NU - 1), a1, b1);
    t2 = tap(u + 5);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_mfma2<CT, PT>(a2, b2, acc, acc2);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_anchor<CT, PT>(a0, b0, t2);
  }
  if (u < NU) conv_chunk_mfma2<CT, PT>(a0, b0, acc, acc2);      // one or two chunks left: sets 0 and 1 hold them
  if (u + 1 < NU) conv_chunk_mfma2<CT, PT>(a1, b1, acc, acc2);
  if constexpr (CT * PT == 1) acc[0][0] += acc2[0][0];
}

// grid = (tiles, parity classes, output-row groups).  PT = position tiles (16 positions each) per wave.
// FZ > 0: fused FeatureNet skip -- the staged tile is computed (1x1 conv of an FZ-channel tensor + bias + nearest
// upsample of the coarser level) instead of copied; the arithmetic is k_skip_up's, so the result is bit-identical to
// running that kernel first and this one on its output.
// One LDS-DMA piece: every lane fetches 16 bytes from its own global address, the wave's 1 KiB lands at LDS byte address
// `lds_dst` (wave-uniform) + lane * 16.  Issued from inline asm on purpose: given the builtin, hipcc orders every later
// LDS read behind the DMA with s_waitcnt vmcnt(0) -- directly in front of the K loop, which serialises the very overlap
// this kernel exists for.  The asm statement is invisible to hipcc's counters; k_conv_a waits for it itself
// (conv_a_wait_dma) before the barrier that publishes the buffer.  M0 is saved and restored around the instruction.
__device__ inline void conv_a_dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ inline void conv_a_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ inline unsigned conv_a_lds_addr(const void *p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)p;
}

// Four waves per SIMD (128 VGPRs) for the one-row-tile instances without the fused skip: they need 129-135 registers when left alone,
// i.e. three workgroups per CU instead of four, and the layers they run (transposed, strided, coarse: a few microseconds of
// work per workgroup between three memory round trips) are bound by exactly that occupancy.  DR_CONV_WAVES3 = the old bound (A/B build).
#ifdef DR_CONV_WAVES3
#define DR_KCONV_MIN_WAVES(CT, FZ) 1
#else
#define DR_KCONV_MIN_WAVES(CT, FZ) ((CT) == 1 && (FZ) == 0 ? 4 : 1)
#endif
// (the body of k_conv and of k_conv_c, its class-loop form: two kernels so that the class loop's code does not sit in the instances that never take it --
// inside one kernel it cost conv9's instance <16,2,4> 25 %, 0.031 -> 0.039 ms, without ever running there)
template <int CI, int CT, int PT, int FZ, bool CL>
__device__ __forceinline__ void conv_tile_body(const ConvArgs &a) {
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
  [[maybe_unused]] const unsigned n_w = (unsigned)NU * CT * 64;
  const int *tp = tapl + sub;
  const unsigned total = (unsigned)NP * C4;
  // PIPELINED PASSES (a.a_wbufs == 2; the planner sets it for the small multi-pass layers: PT = 1, the whole tile in kPipeBatch loads per lane).
  // The coarse UNet levels run 2-4 channel passes of a few microseconds each, and every pass began with a bare round trip to L2 for its tile and
  // its weights (one workgroup per CU or fewer: nobody else covers it).  Here the loads of pass p + 1 -- the tile into registers, the weights by
  // LDS-DMA into the OTHER weight buffer -- are issued before the K loop of pass p and land under it: one exposed round trip per workgroup
  // instead of one per pass.  Same products, same order: bit-identical to the plain loop (DR_CONV_NO_PIPE=1 keeps that one, A/B).
  if constexpr (PT == 1 && CT <= 2 && FZ == 0) {
    if (a.a_wbufs == 2) {
      constexpr int kPipeBatch = 6;
      float4 *wb1 = wl + (size_t)a.nuMax * CT * 64;
      int *tapl2 = reinterpret_cast<int *>(wb1 + (size_t)a.nuMax * CT * 64);  // (the tap table sits behind BOTH weight buffers)
      for (int i = tid; i < NU * TPC; i += kConvThreads) tapl2[i] = a.tapoff[cls.tap_base + i] * CIS;
      const int *tp2 = tapl2 + sub;
      int soff[kPipeBatch], dst[kPipeBatch];  // element offset of this lane's k-th staged float4 inside pass 0's slice (-1: outside the tensor), LDS float index (-1: none)
#pragma unroll
      for (int k = 0; k < kPipeBatch; ++k) {
        const unsigned e = k * kConvThreads + tid;
        const unsigned pos = e / C4, c4 = e - pos * C4;
        const unsigned t = a.magicX ? __umulhi(pos, a.magicX) : pos, x = pos - t * a.TXI;
        const unsigned z = a.magicY ? __umulhi(t, a.magicY) : t, y = t - z * a.TYI;
        const int gz = iz0 + (int)z, gy = iy0 + (int)y, gx = ix0 + (int)x;
        dst[k] = e < total ? (int)(pos * CIS + c4 * 4) : -1;
        soff[k] = (e < total && gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW) ? (int)((((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + c4 * 4) : -1;
      }
      float4 v[kPipeBatch];
      auto issue = [&](int p) {  // weights of pass p -> buffer p & 1 (LDS-DMA), its tile -> registers; nothing is waited for here
        float4 *wb = (p & 1) ? wb1 : wl;
        const float4 *wsrc = a.wpk + cls.w_base + ((size_t)p * NU * a.ctTot + ct0) * 64;
        for (int e = wave; e < NU * CT; e += kConvThreads / 64) {
          const int u = e / CT, ct = e - u * CT;
          conv_a_dma16(wsrc + ((size_t)u * a.ctTot + ct) * 64 + lane, __builtin_amdgcn_readfirstlane(conv_a_lds_addr(wb + (size_t)e * 64)));
        }
#pragma unroll
        for (int k = 0; k < kPipeBatch; ++k) {
          v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (soff[k] >= 0) v[k] = *reinterpret_cast<const float4 *>(a.in + (size_t)soff[k] + (size_t)p * a.pass_stride);
        }
      };
      __builtin_amdgcn_s_setprio(2);
      issue(0);
      for (int p = 0; p < a.npass; ++p) {
#pragma unroll
        for (int k = 0; k < kPipeBatch; ++k)
          if (dst[k] >= 0) *reinterpret_cast<float4 *>(lds + dst[k]) = v[k];
        conv_a_wait_dma();  // (everything in flight belongs to pass p: its tile, consumed just above, and its weights)
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (p + 1 < a.npass) issue(p + 1);  // in flight across the K loop, which touches neither those registers nor that weight buffer
        const float4 *wp = ((p & 1) ? wb1 : wl) + lane;
        conv_kloop_narrow<CT, PT>(tp2, TPC, NU, acc, [&](int toff, int u, float4 (&av)[CT], float4 (&bv)[PT]) { conv_chunk_load<CT, PT>(lds, wp, toff, u, base, av, bv); });
        __syncthreads();  // everybody has left the tile before pass p + 1 overwrites it
      }
      float4 scv[CT], biv[CT];
      conv_load_affine<CT>(a, g, ct0, scv, biv);
      conv_epilogue<CT, PT>(a, cls, acc, scv, biv, wave, j, g, ct0, pz0, py0, px0);
      return;
    }
  }
  // CLASS LOOP (a.class_loop = number of parity classes; single-pass transposed layers: conv11).  With one class per workgroup a layer like stage 2's conv11 is
  // 4800 workgroups whose life is three dependent round trips (weights + tile, barrier, residual operand) around 1.7 us of MFMA work: 0.072 ms, 0.050 of it with the
  // MFMA loop compiled out (profiles/r06_strided_layers_ablation.txt).  Here a workgroup stages its tile ONCE and walks the classes on it, the NEXT class's weights
  // streaming into the other weight buffer (LDS-DMA) under the current class's MFMA loop and epilogue: one exposed round trip per workgroup instead of one per class.
  // Every class's tap table sits in LDS from the start.  Same tile values, same weights, same chunk order per output: bit-identical to the class-per-workgroup launch.
  if constexpr (CL) {
    {
      const int ncls = a.class_loop, tstride = a.nuMax * TPC;
      float4 *wb1 = wl + (size_t)a.nuMax * CT * 64;
      int *tap_all = reinterpret_cast<int *>(wb1 + (size_t)a.nuMax * CT * 64);  // [ncls][nuMax * TPC], behind both weight buffers
      for (int i = tid; i < ncls * tstride; i += kConvThreads) {
        const int c = i / tstride, k = i - c * tstride;
        tap_all[i] = k < a.cls[c].NU * TPC ? a.tapoff[a.cls[c].tap_base + k] * CIS : 0;
      }
      auto issue_w = [&](int c) {  // class c's packed weights -> buffer c & 1
        const unsigned wb = conv_a_lds_addr(wl) + (unsigned)(c & 1) * (unsigned)a.nuMax * CT * 1024u;  // (LDS byte address; an address-space cast of a SELECTED pointer trips hipcc -O2)
        const float4 *wsrc = a.wpk + a.cls[c].w_base + (size_t)ct0 * 64;
        const int n = a.cls[c].NU * CT;
        for (int e = wave; e < n; e += kConvThreads / 64) {
          const int u = e / CT, ct = e - u * CT;
          conv_a_dma16(wsrc + ((size_t)u * a.ctTot + ct) * 64 + lane, __builtin_amdgcn_readfirstlane(wb + (unsigned)e * 1024u));
        }
      };
      __builtin_amdgcn_s_setprio(2);
      issue_w(0);
      constexpr int kStageBatchC = 12;
      for (unsigned e0 = 0; e0 < total; e0 += kConvThreads * kStageBatchC) {
        float4 v[kStageBatchC];
        int dst[kStageBatchC];
#pragma unroll
        for (int k = 0; k < kStageBatchC; ++k) {
          const unsigned e = e0 + k * kConvThreads + tid;
          const unsigned pos = e / C4, c4 = e - pos * C4;
          const unsigned t = a.magicX ? __umulhi(pos, a.magicX) : pos, x = pos - t * a.TXI;
          const unsigned z = a.magicY ? __umulhi(t, a.magicY) : t, y = t - z * a.TYI;
          const int gz = iz0 + (int)z, gy = iy0 + (int)y, gx = ix0 + (int)x;
          v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          dst[k] = e < total ? (int)(pos * CIS + c4 * 4) : -1;
          if (e < total && gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW)
            v[k] = *reinterpret_cast<const float4 *>(a.in + (((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + c4 * 4);
        }
#pragma unroll
        for (int k = 0; k < kStageBatchC; ++k)
          if (dst[k] >= 0) *reinterpret_cast<float4 *>(lds + dst[k]) = v[k];
      }
      float4 scv[CT], biv[CT];
      conv_load_affine<CT>(a, g, ct0, scv, biv);
      for (int c = 0; c < ncls; ++c) {
        conv_a_wait_dma();  // class c's weights have landed (and, first time round, nothing else of this wave is in flight) ...
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();    // ... everybody's have, the tile and the tap tables are written, and every wave has left the buffer class c + 1 goes into
        if (c + 1 < ncls) issue_w(c + 1);
        const ConvClass cc = a.cls[c];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
        const float4 *wc = (c & 1) ? wb1 : wl;
        const int *tpc = tap_all + c * tstride + sub;
        if constexpr (CT * PT <= 2) {
          const float4 *wp = wc + lane;
          conv_kloop_narrow<CT, PT>(tpc, TPC, cc.NU, acc, [&](int toff, int u, float4 (&av)[CT], float4 (&bv)[PT]) { conv_chunk_load<CT, PT>(lds, wp, toff, u, base, av, bv); });
        } else conv_kloop<CT, PT>(lds, wc, tpc, TPC, cc.NU, lane, base, acc);
        conv_epilogue<CT, PT>(a, cc, acc, scv, biv, wave, j, g, ct0, pz0, py0, px0);
      }
      return;
    }
  }
  for (int p = 0; p < a.npass; ++p) {
    // Staging waves get issue priority over co-resident workgroups' K loops: the sooner their loads are in flight the
    // sooner this workgroup can feed the MFMA pipe; the K loop of the neighbour fills the remaining issue slots.
    // (The opposite assignment -- priority to the K loop -- measured 6 % slower.)
    __builtin_amdgcn_s_setprio(2);
    // This pass's packed weights go to LDS by LDS-DMA, requested BEFORE the tile is staged: the round trip to L2 for the
    // weights runs beside the tile's instead of after it, and costs no registers.  (For the coarse layers, whose tiles are
    // small and whose weights are 28-110 KB per pass, the weight fetch was half of the staging step.)  The K loop of the
    // previous pass has left `wl` (barrier at the end of the pass); the pieces have landed at conv_a_wait_dma() below.
    const float4 *wsrc = a.wpk + cls.w_base + ((size_t)p * NU * a.ctTot + ct0) * 64;
#ifndef DR_CONV_NO_WEIGHT_DMA
#ifdef DR_ABL_NO_STAGE
    if (a.npass < 0)
#endif
    for (int e = wave; e < NU * CT; e += kConvThreads / 64) {  // piece e = u * CT + ct: 64 lanes x 16 B, contiguous on both sides
      const int u = e / CT, ct = e - u * CT;
      conv_a_dma16(wsrc + ((size_t)u * a.ctTot + ct) * 64 + lane, __builtin_amdgcn_readfirstlane(conv_a_lds_addr(wl + (size_t)e * 64)));
    }
#endif
    // ---- stage CI channels of the input halo tile into LDS (zero outside the tensor).  Loads are issued in
    // batches of kStageBatch per lane BEFORE the first LDS write so their HBM/L2 latencies overlap. ----
    constexpr int kStageBatch = CT >= 4 ? 6 : 12;  // normally the whole stage: one exposed HBM/L2 latency per pass
    if constexpr (FZ > 0) {
      static_assert(FZ == 0 || (C4 == kConvThreads / 64 && FZ % 4 == 0), "one wave per 4-channel group of the pass");
      // wave w produces channels p*CI + 4w .. +3 of every staged position: its 4 x FZ weights are wave-uniform (scalar
      // registers), a lane reads 4*FZ contiguous bytes of x (64 lanes = one contiguous run) and writes one float4.
      const int q = __builtin_amdgcn_readfirstlane(p * C4 + wave);
      float wr[4][FZ];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < FZ; ++c) wr[r][c] = a.fz_w[(4 * q + r) * FZ + c];
      const float4 fb = *reinterpret_cast<const float4 *>(a.fz_b + 4 * q);
#ifndef DR_FZ_BATCH
#define DR_FZ_BATCH 4
#endif
      constexpr int kFB = DR_FZ_BATCH;
      const int Hc = a.inH >> 1, Wc = a.inW >> 1;
      for (int p0 = 0; p0 < NP; p0 += 64 * kFB) {
        float4 xv[kFB][FZ / 4], up[kFB];
        int dst[kFB];
        bool in[kFB];
#pragma unroll
        for (int k = 0; k < kFB; ++k) {
          const unsigned pos = p0 + k * 64 + lane;
          const unsigned t = a.magicX ? __umulhi(pos, a.magicX) : pos, x = pos - t * a.TXI;
          const unsigned z = a.magicY ? __umulhi(t, a.magicY) : t, y = t - z * a.TYI;
          const int gz = iz0 + (int)z, gy = iy0 + (int)y, gx = ix0 + (int)x;
          dst[k] = (int)pos < NP ? (int)(pos * CIS + wave * 4) : -1;
          in[k] = (int)pos < NP && gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW;
          up[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int c = 0; c < FZ / 4; ++c) xv[k][c] = up[k];
          if (in[k]) {
            const float *xp = a.fz_x + (((size_t)gz * a.inH + gy) * a.inW + gx) * FZ;
#pragma unroll
            for (int c = 0; c < FZ / 4; ++c) xv[k][c] = *reinterpret_cast<const float4 *>(xp + 4 * c);
            up[k] = *reinterpret_cast<const float4 *>(a.fz_coarse + (((size_t)gz * Hc + (gy >> 1)) * Wc + (gx >> 1)) * a.inC + 4 * q);
          }
        }
#pragma unroll
        for (int k = 0; k < kFB; ++k) {
          float acc4[4] = {0.f, 0.f, 0.f, 0.f}, xi[FZ];
#pragma unroll
          for (int c = 0; c < FZ / 4; ++c) { xi[4 * c] = xv[k][c].x; xi[4 * c + 1] = xv[k][c].y; xi[4 * c + 2] = xv[k][c].z; xi[4 * c + 3] = xv[k][c].w; }
#pragma unroll
          for (int sft = 0; sft < 4; ++sft)
#pragma unroll
            for (int gq = 0; gq < FZ / 4; ++gq)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc4[r] = __builtin_fmaf(wr[r][4 * gq + sft], xi[4 * gq + sft], acc4[r]);
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);  // outside the tensor: the 3x3 layer's zero padding
          if (in[k]) { o.x = (acc4[0] + fb.x) + up[k].x; o.y = (acc4[1] + fb.y) + up[k].y; o.z = (acc4[2] + fb.z) + up[k].z; o.w = (acc4[3] + fb.w) + up[k].w; }
          if (dst[k] >= 0) *reinterpret_cast<float4 *>(lds + dst[k]) = o;
        }
      }
    } else
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
          v[k] = *reinterpret_cast<const float4 *>(a.in + (((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + (size_t)p * a.pass_stride + c4 * 4);
      }
#pragma unroll
      for (int k = 0; k < kStageBatch; ++k)
        if (dst[k] >= 0) *reinterpret_cast<float4 *>(lds + dst[k]) = v[k];
    }
#ifndef DR_CONV_NO_WEIGHT_DMA
    conv_a_wait_dma();
#else
    {  // (A/B build: the weights through registers, after the tile)
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
#endif
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
#ifndef DR_ABL_NO_KLOOP
#ifdef DR_NO_NARROW_KLOOP  // A/B build
    if constexpr (false) {
#else
    if constexpr (CT * PT <= 2) {
#endif
      const float4 *wp = wl + lane;
      conv_kloop_narrow<CT, PT>(tp, TPC, NU, acc, [&](int toff, int u, float4 (&av)[CT], float4 (&bv)[PT]) { conv_chunk_load<CT, PT>(lds, wp, toff, u, base, av, bv); });
    } else conv_kloop<CT, PT>(lds, wl, tp, TPC, NU, lane, base, acc);
#endif
    __syncthreads();
  }

  float4 scv[CT], biv[CT];
  conv_load_affine<CT>(a, g, ct0, scv, biv);
  conv_epilogue<CT, PT>(a, cls, acc, scv, biv, wave, j, g, ct0, pz0, py0, px0);
}
template <int CI, int CT, int PT, int FZ = 0>
__global__ __launch_bounds__(kConvThreads, DR_KCONV_MIN_WAVES(CT, FZ)) void k_conv(const ConvArgs a) {
  conv_tile_body<CI, CT, PT, FZ, false>(a);
}
// k_conv_c: k_conv's CLASS LOOP form (a.class_loop parity classes per workgroup, grid.y = 1), under k_conv's register bound
template <int CI, int CT, int PT>
__global__ __launch_bounds__(kConvThreads, DR_KCONV_MIN_WAVES(CT, 0)) void k_conv_c(const ConvArgs a) {
  conv_tile_body<CI, CT, PT, 0, true>(a);
}
inline bool conv_c_instance_exists(int ci, int ct, int pt) { return (ci == 8 || ci == 16) && (ct == 1 || ct == 2) && (pt == 1 || pt == 4); }

#ifdef DR_PARITY_HOOKS  // the bf16 x 3 precision mode is not fp32 arithmetic and never the product's: its kernel is built into the parity library only
#include "conv_bf3.h"  // k_conv_b: k_conv on the bf16 matrix cores with three-term split operands (DR_CONV_BF16X3=1)
#endif

// ------------------------------------------------------------------------------------------------
// k_conv_a: the same implicit GEMM as a PERSISTENT workgroup whose staging is asynchronous.
//
// k_conv's phases are additive (r2 ablation at s2.conv0: K loop 61 %, staging 24 %, epilogue + launch 11 %; two
// co-resident workgroups run in phase, so they do not hide each other's staging).  Here a workgroup of 8 waves walks a
// list of (tile, channel pass) units; while the MFMAs of unit i run out of LDS buffer i & 1, the halo tile (and, for
// multi-pass layers, the packed weights) of unit i + 1 stream into the other buffer with global_load_lds_dwordx4 --
// no staging registers, no ds_write pass, one barrier per unit.  The LDS-DMA destination is wave-uniform base +
// lane * 16, so the tile is stored unpadded; bank conflicts of the operand reads are avoided by a slot permutation
// applied on the SOURCE side (each lane fetches the element that belongs in its slot) and on the read side
// (tools/ubench/glds_probe.hip: conflict-free for position strides 1 and 2, which is what the layers use).
constexpr int kConvAThreads = 512;

template <int CI>
__host__ __device__ inline int conv_a_unit(int pos, int c4) {  // (position, channel group) -> 16-byte slot
  if constexpr (CI == 4) return pos;
  else if constexpr (CI == 8) return ((pos ^ ((pos >> 3) & 1)) << 1) | c4;
  else return ((pos ^ ((pos >> 3) & 1)) << 2) | (c4 ^ ((pos >> 1) & 3));
}
template <int CI>
__host__ __device__ inline void conv_a_slot(int s, int &pos, int &c4) {  // inverse of conv_a_unit (both xors are involutions)
  if constexpr (CI == 4) { pos = s; c4 = 0; }
  else if constexpr (CI == 8) { const int pp = s >> 1; pos = pp ^ ((pp >> 3) & 1); c4 = s & 1; }
  else { const int pp = s >> 2; pos = pp ^ ((pp >> 3) & 1); c4 = (s & 3) ^ ((pos >> 1) & 3); }
}

template <int CI, int CT>
__device__ inline void conv_a_issue(const ConvArgs &a, float4 *tile, float4 *wbuf, bool with_weights, int NP, int NU, int p, int ct0, int w_base,
                                    int iz0, int iy0, int ix0, int wave, int lane) {
  for (int s0 = wave * 64; s0 < a.a_slots; s0 += kConvAThreads) {  // s0 is wave-uniform
    int pos, c4;
    conv_a_slot<CI>(s0 + lane, pos, c4);
    const unsigned t = a.magicX ? __umulhi((unsigned)pos, a.magicX) : (unsigned)pos, x = pos - t * a.TXI;
    const unsigned z = a.magicY ? __umulhi(t, a.magicY) : t, y = t - z * a.TYI;
    const int gz = iz0 + (int)z, gy = iy0 + (int)y, gx = ix0 + (int)x;
    const float *src = a.zero16;
    if (pos < NP && gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW)
      src = a.in + (((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + (size_t)p * a.pass_stride + c4 * 4;
    conv_a_dma16(src, __builtin_amdgcn_readfirstlane(conv_a_lds_addr(tile + s0)));
  }
  if (with_weights) {  // packed weights of this pass and row group: [u][CT][64] float4, lane-linear as they are in memory
    const float4 *wsrc = a.wpk + w_base + ((size_t)p * NU * a.ctTot + ct0) * 64;
    for (int e0 = wave * 64; e0 < NU * CT * 64; e0 += kConvAThreads) {
      const int u = e0 / (CT * 64), r = e0 - u * (CT * 64);
      conv_a_dma16(wsrc + (size_t)u * a.ctTot * 64 + r + lane, __builtin_amdgcn_readfirstlane(conv_a_lds_addr(wbuf + e0)));
    }
  }
}

template <int CI, int CT, int PT>
__device__ inline void conv_a_load(const float4 *tile, const float4 *wp, int toff, int u, int c4, const int (&bpos)[PT], float4 (&av)[CT], float4 (&bv)[PT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) av[ct] = wp[(u * CT + ct) * 64];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) bv[pt] = tile[conv_a_unit<CI>(bpos[pt] + toff, c4)];
}
template <int CI, int CT, int PT>
__device__ inline void conv_a_kloop(const float4 *tile, const float4 *wl, const int *tp, int TPC, int NU, int lane, int c4, const int (&bpos)[PT],
                                    floatx4 (&acc)[CT][PT]) {
  const float4 *wp = wl + lane;
  float4 a0[CT], b0[PT], a1[CT], b1[PT];
  int tA = tp[0], tB = tp[min(1, NU - 1) * TPC];
  conv_a_load<CI, CT, PT>(tile, wp, tA, 0, c4, bpos, a0, b0);
  int u = 0;
  for (; u + 1 < NU; u += 2) {
    conv_a_load<CI, CT, PT>(tile, wp, tB, u + 1, c4, bpos, a1, b1);
    tA = tp[min(u + 2, NU - 1) * TPC];
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_mfma<CT, PT>(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_anchor<CT, PT>(a1, b1, tA);
    conv_a_load<CI, CT, PT>(tile, wp, tA, min(u + 2, NU - 1), c4, bpos, a0, b0);
    tB = tp[min(u + 3, NU - 1) * TPC];
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_mfma<CT, PT>(a1, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    conv_chunk_anchor<CT, PT>(a0, b0, tB);
  }
  if (u < NU) conv_chunk_mfma<CT, PT>(a0, b0, acc);
}

// grid = (persistent workgroups (multiple of 8), 1, output-row groups); 8 waves, PT position tiles per wave.
template <int CI, int CT, int PT>
__global__ __launch_bounds__(kConvAThreads) void k_conv_a(const ConvArgs a) {
  extern __shared__ float4 lds4[];
  constexpr int TPC = 16 / CI;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, g = lane >> 4;
  const ConvClass cls = a.cls[0];
  const int ct0 = blockIdx.z * CT;
  const int NP = a.TZI * a.TYI * a.TXI, NU = cls.NU, n_w = NU * CT * 64;

  // LDS: [tile 0][tile 1][weights 0][weights 1 (multi-pass layers)][tap table]
  float4 *tile0 = lds4, *tile1 = lds4 + a.a_slots;
  float4 *wb0 = lds4 + 2 * (size_t)a.a_slots;  // weight buffer of pass p: wb0 + (p % a_wbufs) * n_w
  int *tapl = reinterpret_cast<int *>(wb0 + (size_t)a.a_wbufs * n_w);
  for (int i = tid; i < NU * TPC; i += kConvAThreads) tapl[i] = a.tapoff[cls.tap_base + i];
  const int *tp = tapl + (4 * g) / CI;
  const int c4 = ((4 * g) % CI) / 4;

  int bpos[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int tau = wave * PT + pt;
    const int xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
    bpos[pt] = ((zt * a.sz) * a.TYI + yt * a.sy) * a.TXI + (xt * 16 + j) * a.sx;
  }

  // this workgroup's tiles: XCD k (= blockIdx.x % 8, own L2) owns the k-th contiguous range of tiles; its workgroups
  // take them round-robin, so at any time an XCD works on neighbouring tiles whose halos overlap in its L2
  const int ntiles = a.tilesD * a.tilesH * a.tilesW, per_xcd = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, nw = gridDim.x >> 3;
  const int t_lo = xcd * per_xcd, t_hi = min(ntiles, t_lo + per_xcd);
  const int my_tiles = t_lo + wi < t_hi ? (t_hi - t_lo - wi + nw - 1) / nw : 0;
  const int n_units = my_tiles * a.npass;
  auto tile_origin = [&](int k, int &pz0, int &py0, int &px0) {
    int b = t_lo + wi + k * nw;
    const int tw = b % a.tilesW;
    b /= a.tilesW;
    pz0 = (b / a.tilesH) * a.TZ; py0 = (b % a.tilesH) * a.TY; px0 = tw * a.TXT * 16;
  };
  auto issue = [&](int unit) {
    const int k = unit / a.npass, p = unit - k * a.npass;
    int pz0, py0, px0;
    tile_origin(k, pz0, py0, px0);
    // weight buffer = pass % buffers: when every pass has its own buffer it is fetched once per workgroup
    conv_a_issue<CI, CT>(a, (unit & 1) ? tile1 : tile0, wb0 + (size_t)(p % a.a_wbufs) * n_w, unit < a.npass || a.npass > a.a_wbufs, NP, NU, p, ct0, cls.w_base,
                         pz0 * a.sz - a.pz, py0 * a.sy - a.py, px0 * a.sx - a.px, wave, lane);
  };

  floatx4 acc[CT][PT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};

  float4 scv[CT], biv[CT];  // folded BatchNorm of this lane's rows: fetched once, not per tile
  conv_load_affine<CT>(a, g, ct0, scv, biv);
  if (n_units > 0) issue(0);
  int done_tile = -1;  // tile whose accumulators are complete and not yet written
  for (int i = 0; i < n_units; ++i) {
    conv_a_wait_dma();  // this wave's pieces of unit i have landed ...
    __syncthreads();    // ... and so have everybody else's; every wave is done reading the other buffer
    // The next unit's DMA goes out BEFORE the previous tile's epilogue when that epilogue only stores (no residual operand): the round trip to L2 /
    // HBM then runs beside the epilogue as well as the K loop (a stride-2 layer's K loop is 0.7 us against a 2-3 us round trip).  With a residual
    // operand the order stays epilogue first: the DMA is invisible to hipcc's counters, and the vmcnt it puts behind the residual loads would then
    // wait for the (older) DMA pieces too.
#ifndef DR_ABL_NO_STAGE
#ifndef DR_CONV_A_ISSUE_LATE  // (A/B build: always after the epilogue, round 3's order)
    const bool early = a.add_mode == 0;
#else
    const bool early = false;
#endif
    if (early && i + 1 < n_units) issue(i + 1);
#endif
#ifdef DR_ABL_NO_EPI
    if (done_tile >= 0 && a.npass < 0) {
#else
    if (done_tile >= 0) {  // epilogue of the previous tile: its stores retire under the K loop below
#endif
      int pz0, py0, px0;
      tile_origin(done_tile, pz0, py0, px0);
      if (a.add_mode) conv_epilogue<CT, PT, true>(a, cls, acc, scv, biv, wave, j, g, ct0, pz0, py0, px0);
      else conv_epilogue<CT, PT, false>(a, cls, acc, scv, biv, wave, j, g, ct0, pz0, py0, px0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
      done_tile = -1;
    }
#ifndef DR_ABL_NO_STAGE
    if (!early && i + 1 < n_units) issue(i + 1);
#endif
#ifndef DR_ABL_NO_KLOOP
#ifdef DR_NO_NARROW_KLOOP
    if constexpr (false) {
#else
    if constexpr (CT * PT <= 2) {
#endif
      const float4 *tile = (i & 1) ? tile1 : tile0, *wp = wb0 + (size_t)((i % a.npass) % a.a_wbufs) * n_w + lane;
      conv_kloop_narrow<CT, PT>(tp, TPC, NU, acc, [&](int toff, int u, float4 (&av)[CT], float4 (&bv)[PT]) { conv_a_load<CI, CT, PT>(tile, wp, toff, u, c4, bpos, av, bv); });
    } else conv_a_kloop<CI, CT, PT>((i & 1) ? tile1 : tile0, wb0 + (size_t)((i % a.npass) % a.a_wbufs) * n_w, tp, TPC, NU, lane, c4, bpos, acc);
#endif
    if ((i + 1) % a.npass == 0) done_tile = i / a.npass;
  }
  if (done_tile >= 0) {
    int pz0, py0, px0;
    tile_origin(done_tile, pz0, py0, px0);
    conv_epilogue<CT, PT>(a, cls, acc, scv, biv, wave, j, g, ct0, pz0, py0, px0);
  }
}

}  // namespace dr
#include "conv_wino.h"   // k_conv_w: k_conv with the y axis of a 3-tap stride-1 layer in Winograd F(2,3) form (two thirds of the MFMAs)
namespace dr {
template <int CI> __host__ __device__ inline int conv_a_unit(int pos, int c4);
}
#include "conv_march.h"  // k_conv_m: marching producer/consumer kernel for the stride-1 3x3 / 3x3x3 layers (and their Winograd form)
namespace dr {

// ------------------------------------------------------------------------------------------------
// Host side: logical layer description -> packed weights, tap table, tile plan, launches.

enum ConvMode { CONV_NORMAL = 0, CONV_XPAIR = 1, CONV_X8 = 2 };

struct ConvLayer {  // logical description (torch semantics)
  int Cin = 0, Cout = 0;
  int kd = 1, kh = 1, kw = 1;
  int sd = 1, sh = 1, sw = 1;
  bool transposed = false;          // ConvTranspose3d(k=3, pad=1, output_padding = stride-1)
  int up2 = 0;                      // 1 + py: Conv2d(k=3, pad=1) applied to the nearest x2 upsampling (in H and W) of the input,
                                    // which is given at HALF resolution and never upsampled in memory; this launch produces the output rows 2Y + py.  Output
                                    // pixel (2Y+py, 2X+px) reads 2 x 2 input pixels, the kernel rows / columns that fall on the same input
                                    // pixel summed (weight (Cout,Cin,1,3,3)); both px are rows of the launch (2 * Cout rows)
  const float *weight = nullptr;    // (Cout,Cin,kd,kh,kw) or transposed (Cin,Cout,kd,kh,kw)
  std::vector<float> scale, bias;   // per Cout (folded BN / conv bias); empty -> 1 / 0
  bool relu = false;
  int out_pad = 0;                  // the output tensor carries a border of this many pixels in H and W (`out` points at its first interior pixel)
};

struct ConvFuse {  // FeatureNet skip pair fused into the staging step of the layer that consumes it (device pointers)
  const float *x, *w, *b, *coarse;
  int cin;  // channels of x (8)
};

struct ConvLaunch {
  ConvArgs args;
  int ci, ct, pt, fz = 0;
  int async = 0;  // 1: k_conv_a (persistent, LDS-DMA staged, 512 threads); 2: k_conv_m (marching, 8 consumer + 2 producer waves)
  MarchArgs march{};
  int nup = 0;    // k_conv_m: K chunks per input plane
  int ncw = 8;    // k_conv_m: consumer waves (8 or 12)
  int bf3 = 0;    // 1: k_conv_b (bf16 x 3 operands; wpk holds hi / lo bf16 fragments, 32-wide K chunks)
  dim3 grid;
  size_t lds_bytes;
  double flops;  // useful (algorithmic) flops of this launch
};

struct DimTaps {               // per-axis decomposition of one parity class
  std::vector<int> t, off;     // kernel index, input offset (>= 0)
  int s = 1, p = 0, om = 1, oo = 0, npos = 0;
  int par = -1;                // transposed stride-2 axis split into classes: the output parity this class produces (-1: not split)
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
// A strided transposed axis comes in two forms.  DENSE: both parities are output ROWS of one class (offsets {0,1}, zero weight
// where parity 0 does not use offset 1) -- one class, but a quarter of the products per axis are multiplications by zero.
// SPLIT: one class per parity (parity 0: offset 0 only; parity 1: offsets 0 and 1) -- no zero products, half the rows.
// An up2 axis (ConvLayer::up2): output 2m + par of the 3-tap kernel over the upsampled signal reads input {m-1, m} (par 0: kernel
// index 0 | indices 1 and 2 summed) or {m, m+1} (par 1: indices 0 and 1 summed | index 2); offsets are relative to m - 1 (pad 1).
inline std::vector<DimTaps> axis_classes_up2(int in_size, bool split) {
  std::vector<DimTaps> r;
  if (!split) { DimTaps d; d.t = {0, 1, 2}; d.off = {0, 1, 2}; d.p = 1; d.om = 2; d.npos = in_size; r.push_back(d); return r; }
  DimTaps e; e.t = {0, 1}; e.off = {0, 1}; e.p = 1; e.om = 2; e.oo = 0; e.npos = in_size; e.par = 0; r.push_back(e);
  DimTaps o; o.t = {1, 2}; o.off = {1, 2}; o.p = 1; o.om = 2; o.oo = 1; o.npos = in_size; o.par = 1; r.push_back(o);
  return r;
}
// kernel indices that (parity, input offset) of an up2 axis sums; returns the count (0: the pair carries no weight)
inline int up2_kernel_set(int parity, int off, int (&k)[2]) {
  if (parity == 0) { if (off == 0) { k[0] = 0; return 1; } if (off == 1) { k[0] = 1; k[1] = 2; return 2; } return 0; }
  if (off == 1) { k[0] = 0; k[1] = 1; return 2; }
  if (off == 2) { k[0] = 2; return 1; }
  return 0;
}
// Taps that can only ever read padding are dropped: on an axis of extent 1 a 3-tap stride-1 kernel touches the tensor through its centre tap
// alone (stage 3's conv6 runs on a depth of 1: two thirds of its products were multiplications by staged zeros), a stride-2 kernel on an axis of
// extent 2 through two of its three taps (conv5), a transposed stride-2 axis of extent 1 never reads x[m + 1] (conv7).  The remaining offsets are
// shifted so that the smallest is 0 (the padding shrinks with them): the halo a tile stages shrinks too.
inline void prune_taps(DimTaps &d, int in_size) {
  std::vector<int> t, off;
  for (size_t i = 0; i < d.t.size(); ++i) {
    bool used = false;
    for (int pos = 0; pos < d.npos && !used; ++pos) { const int x = pos * d.s - d.p + d.off[i]; used = x >= 0 && x < in_size; }
    if (used) { t.push_back(d.t[i]); off.push_back(d.off[i]); }
  }
  if (t.empty() || t.size() == d.t.size()) return;
  const int mo = *std::min_element(off.begin(), off.end());
  for (int &o : off) o -= mo;
  d.p -= mo;
  d.t = t; d.off = off;
}
inline std::vector<DimTaps> axis_classes_unpruned(int k, int s, bool transposed, int in_size, bool split);
// (prune: the planner asks for it on the DEPTH axis only -- the in-plane axes of every layer are far longer than their kernels, and the row march /
// Winograd forms are written for exactly three y taps)
inline std::vector<DimTaps> axis_classes(int k, int s, bool transposed, int in_size, bool split = false, bool prune = false) {
  std::vector<DimTaps> r = axis_classes_unpruned(k, s, transposed, in_size, split);
  if (prune && !hook_env("DR_CONV_NO_TAP_PRUNE"))
    for (DimTaps &d : r) prune_taps(d, in_size);
  return r;
}
inline std::vector<DimTaps> axis_classes_unpruned(int k, int s, bool transposed, int in_size, bool split) {
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
  } else if (!split) {
    // dense parity form: input offsets {0,1}; which kernel index a (parity, offset) pair selects is resolved when the
    // weights are packed (parity_kernel_index), the two parities of this axis become separate output ROWS
    DimTaps d; d.t = {0, 1}; d.off = {0, 1}; d.om = 2; d.oo = 0; d.npos = in_size; r.push_back(d);
  } else {
    DimTaps e; e.t = {0}; e.off = {0}; e.om = 2; e.oo = 0; e.npos = in_size; e.par = 0; r.push_back(e);
    DimTaps o; o.t = {0, 1}; o.off = {0, 1}; o.om = 2; o.oo = 1; o.npos = in_size; o.par = 1; r.push_back(o);
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
  int *err_flag = nullptr;  // where k_conv_m reports a wait that gave up (device-visible memory owned by the caller)
  bool host_only = false;   // planning without a device (tests/cpp/march_emul.hip): the "uploads" stay in host memory
  template <class T>
  T *upload(const std::vector<T> &h) {
    if (host_only) {
      T *d = static_cast<T *>(malloc(std::max<size_t>(h.size() * sizeof(T), 16)));
      memcpy(d, h.data(), h.size() * sizeof(T));
      ptrs.push_back(d);
      return d;
    }
    T *d = dalloc<T>(h.size());
    DR_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    ptrs.push_back(d);
    return d;
  }
  ~DeviceArena() { for (void *p : ptrs) { if (host_only) free(p); else (void)hipFree(p); } }
};

// Measured-best plans for known layer shapes (tools/tune_conv.sh -> conv_tuned.h); anything else uses the cost model.
struct ConvTuned {
  int Cin, Cout, kd, kh, kw, sd, sh, sw, transposed, mode, inD, inH, inW;  // layer signature
  int ci, ct, pt, tz, ty, txt;                                              // plan: channel pass, row tiles, position tiles per wave, tile shape (txt in 16s)
  int async_;                                                               // 1: the persistent LDS-DMA kernel (k_conv_a)
};
#include "conv_tuned.h"

inline bool conv_instance_exists(int ci, int ct) {
  return (ci == 4 && ct == 1) || ((ci == 8 || ci == 16) && (ct == 1 || ct == 2 || ct == 4));
}
inline bool conv_a_instance_exists(int ci, int ct, int pt) {  // PT = 1 (round 4): 128-position tiles, for the layers whose halo (stride 2, 5 x 5) does not fit twice at 256
  return (pt == 1 || pt == 2 || pt == 4) && ((ci == 4 && ct == 1 && pt != 1) || ((ci == 8 || ci == 16) && (ct == 1 || ct == 2)));
}
// k_conv_m instances (CI, NUP, CT, PT, consumer waves).  3-D layers: NUP = 6 (8 channels, XPAIR), 12 (16 channels, XPAIR), 9 (16 channels);
// row march of 2-D layers: NUP = 2 (8 channels, XPAIR), 4 (16 channels, XPAIR), 3 (16 channels).  CT = 2 with PT = 4 needs more than the
// 168 registers three waves per SIMD leave; twelve consumer waves leave 128 each.
#define DR_MARCH_INSTANCES(X)                                                                                         \
  X(8, 6, 1, 2, 8) X(8, 6, 1, 4, 8) X(8, 6, 1, 1, 12) X(8, 6, 1, 2, 12)                                               \
  X(16, 12, 1, 2, 8) X(16, 12, 1, 4, 8) X(16, 12, 1, 1, 12) X(16, 12, 1, 2, 12)                                       \
  X(16, 9, 1, 2, 8) X(16, 9, 1, 4, 8) X(16, 9, 1, 1, 12) X(16, 9, 1, 2, 12) X(16, 9, 2, 2, 8)                         \
  X(8, 2, 1, 1, 8) X(8, 2, 1, 2, 8) X(8, 2, 1, 4, 8) X(8, 2, 1, 1, 10) X(8, 2, 1, 2, 10)                              \
  X(16, 4, 1, 1, 8) X(16, 4, 1, 2, 8) X(16, 4, 1, 4, 8) X(16, 4, 1, 1, 10) X(16, 4, 1, 2, 10)                         \
  X(16, 3, 1, 1, 8) X(16, 3, 1, 2, 8) X(16, 3, 1, 4, 8) X(16, 3, 1, 1, 10) X(16, 3, 1, 2, 10)                         \
  X(16, 3, 2, 1, 8) X(16, 3, 2, 2, 8) X(16, 3, 2, 1, 10) X(16, 3, 2, 2, 10)
// the Winograd form of the 3-D instances (march_consumer_w): one position tile (16 x by a row pair) per wave
#define DR_MARCH_W_INSTANCES(X) X(8, 6, 1, 1, 8) X(16, 12, 1, 1, 8) X(16, 9, 1, 1, 8)  // (twelve consumer waves leave 128 registers: the two operand sets spill)
inline bool conv_m_instance_exists(int ci, int nup, int ct, int pt, int ncw, bool wino = false) {
#define DR_X(CI_, NUP_, CT_, PT_, NCW_) if (ci == CI_ && nup == NUP_ && ct == CT_ && pt == PT_ && ncw == NCW_) return true;
  if (wino) { DR_MARCH_W_INSTANCES(DR_X) return false; }
  DR_MARCH_INSTANCES(DR_X)
#undef DR_X
  return false;
}
// DR_CONV_MARCH: 0 = never plan k_conv_m, 1 = rank it with the other families (default), 2 = prefer it wherever it applies (A/B hook)
inline int conv_march_policy() {
  if (const char *e = getenv("DR_CONV_MARCH")) return atoi(e);
  return 1;
}
// DR_CONV_ROWMARCH: the same for the row march of 2-D layers (0 = never, 1 = ranked, 2 = preferred)
inline int conv_rowmarch_policy() {
  if (const char *e = getenv("DR_CONV_ROWMARCH")) return atoi(e);
  return 1;
}
struct MarchShape {  // derived geometry of a k_conv_m candidate
  int nup, npi, npo, ns, tyi, txi, np, ps, r, ncw;
  size_t wbytes, lds_bytes;
  long long steps;
  int grid;
  bool ok;
};
// rm (row march of a 2-D layer): KZ is the layer's kd (1), ntp the x taps of ONE row, nPD the images, nPH the rows; ty must be 1.
// wino (march_consumer_w; KZ = 3 only): ty counts row PAIRS, nPH the pairs of the layer, ntp the taps of one plane as the DIRECT form has them (3 rows x the x taps)
inline MarchShape march_shape(int KZ, int ntp, int Cin, int ci, int ct, int pt, int ty, int txt, int SX, int exy, int exx, int nPD, int nPH, int nPW, int CTtot,
                              bool rm = false, bool wino = false) {
  MarchShape m{};
  const int tpc = 16 / ci;
  if (Cin % ci || ntp % tpc || CTtot % ct || (ty * txt) % pt || (rm && (ty != 1 || KZ != 1)) || (wino && (KZ != 3 || rm || (ntp / 3) % tpc))) return m;
  m.ncw = ty * txt / pt;  // one wave per PT position tiles
  m.nup = ntp / tpc;      // (Winograd form: the same count -- three raw kernel rows per chunk of x taps)
  if (!conv_m_instance_exists(ci, m.nup, ct, pt, m.ncw, wino)) return m;
  const int npass = Cin / ci, kz = rm ? 3 : KZ;
  if (npass > 2) return m;
  m.npi = KZ == 1 ? npass : 1; m.npo = KZ == 1 ? 1 : npass; m.ns = kz * m.npi;
  m.tyi = rm ? 1 : (wino ? 2 * ty + 2 : ty - 1 + exy); m.txi = (txt * 16 - 1) * SX + exx; m.np = m.tyi * m.txi;
  if (m.np >= 65536) return m;
  m.ps = cdiv(((m.np + 15) & ~15) * (ci / 4), 128) * 128;
  if (m.ps / 128 > kMarchMaxIt) return m;
  m.wbytes = (size_t)m.ns * m.nup * ct * 1024;
  const size_t fixed = m.wbytes + kMarchFlagInts * 4;
  if (fixed >= kConvMaxLds) return m;
  // ring: the planes a step reads, plus what the producers may fetch ahead (rows are small: two whole rows ahead when they fit)
  const int rmin = kz == 3 ? 3 * m.npi : m.npi + 1, rmax = rm ? 5 * m.npi : (kz == 3 ? 4 : 2 * m.npi + 2);
  m.r = (int)std::min<size_t>(rmax, (kConvMaxLds - fixed) / ((size_t)m.ps * 16));
  if (m.r < rmin) return m;
  m.lds_bytes = (size_t)m.r * m.ps * 16 + fixed;
  m.steps = (long long)(rm ? 1 : cdiv(nPH, ty)) * cdiv(nPW, txt * 16) * nPD * (rm ? nPH : 1);
  const int split = CTtot / ct;
  m.grid = 8 * cdiv((int)std::min<long long>(m.steps, std::max(8, 256 / split)), 8);
  m.ok = true;
  return m;
}
// k_conv_w instances (CI, CT, PT): 16-channel passes with one or two row tiles, 8-channel passes (Cin = 8) with one
inline bool conv_w_instance_exists(int ci, int ct, int pt) {
  return (pt == 1 || pt == 2) && ((ci == 16 && (ct == 1 || ct == 2)) || (ci == 8 && ct == 1));  // (PT = 4 needs more than 256 registers: 64 accumulators + two operand sets of 80)
}
// DR_CONV_WINO: 0 = never plan k_conv_w, 1 = rank it with the other families (default), 2 = prefer it wherever it applies (A/B hook, tests)
inline int conv_wino_policy() {
  if (const char *e = getenv("DR_CONV_WINO")) return atoi(e);
  return 1;
}
// Winograd F(2,3) weight transform along one 3-tap axis: point p of (g0, g1, g2)
inline double conv_wino_g(int p, int k) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  return G[p][k];
}
inline size_t conv_a_slots(int np, int ci) { return (size_t)cdiv(((np + 1) & ~1) * (ci / 4), 512) * 512; }
// Form of a stride-2 transposed layer (see axis_classes): 0 = every strided axis dense (8 * Cout rows, 27 of 64 products useful),
// 1 = x dense, z and y split (2 * Cout rows in 4 classes, 3 of 4 useful), 2 = every axis split (Cout rows in 8 classes, all
// useful).  Default by measurement at 640x480 (profiles/r03_experiments.txt): conv11 (Cout 8) 0.076 / 0.073 / 0.126 ms, conv9 (Cout 16)
// 0.039 / 0.030 / 0.034, conv7 (Cout 32) 0.0265 / 0.0233 / 0.0221 for forms 0 / 1 / 2.  DR_DECONV_FORM overrides (A/B hook).
inline int conv_deconv_form(int Cout) {
  if (const char *e = hook_env("DR_DECONV_FORM")) return std::max(0, std::min(2, atoi(e)));
  return Cout >= 32 ? 2 : 1;
}
// which kernel family the planner may use: 0 = k_conv only, 1 = k_conv_a only (falls back to k_conv when no async plan
// fits), 2 = both, ranked together.  DR_CONV_ASYNC overrides (A/B hook).
// Opt-in precision mode (DR_CONV_BF16X3=1): every layer with Cin % 8 == 0 runs on k_conv_b -- k_conv's data flow on the bf16 matrix cores
// with both operands split into two bf16 terms and the three leading products accumulated in fp32 (conv_bf3.h has the numerics).
inline int conv_bf3_policy() {
  const char *e = hook_env("DR_CONV_BF16X3");
  return e && atoi(e) > 0 ? 1 : 0;
}
inline unsigned short bf16_rne(float f) {  // round to nearest even, as v_cvt_pk_bf16_f32 does
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
inline float bf16_value(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline int conv_async_policy() {
  if (const char *e = getenv("DR_CONV_ASYNC")) return atoi(e);
  return 2;
}

// in_split: 0 = the input is one (D,H,W,inC) tensor; 16 = it is stored as inC / 16 consecutive (D,H,W,16) sub-tensors (only 16-channel passes apply;
// a tuned row of such a layer carries mode + 16: plans measured in one layout are never applied to the other)
inline ConvPlanOut plan_conv(const ConvLayer &L, ConvMode mode, const float *in, int inD, int inH, int inW, int inC,
                             float *out, const float *add, int add_mode, DeviceArena &arena, int rank = 0, const ConvFuse *fz = nullptr, int in_split = 0) {
  // rank: which candidate of the cost model's ranking to build (0 = its choice); used by the engine's autotuner
  if (L.Cin % 4 != 0 || inC < L.Cin) fail(DR_ERR_ARG, "plan_conv: Cin=%d must be a multiple of 4 (tensor C=%d)", L.Cin, inC);
  const int form = L.transposed ? conv_deconv_form(L.Cout) : 0;
  if (L.up2 && (L.up2 > 2 || L.transposed || L.kd != 1 || L.kh != 3 || L.kw != 3 || L.sd != 1 || L.sh != 1 || L.sw != 1 || mode != CONV_NORMAL))
    fail(DR_ERR_ARG, "plan_conv: up2 is a plain 3x3 stride-1 2-D layer over the upsampled input");
  auto cz = axis_classes(L.kd, L.sd, L.transposed, inD, form >= 1, true);
  auto cy = axis_classes(L.kh, L.sh, L.transposed, inH, form >= 1);
  auto cx = axis_classes(L.kw, L.sw, L.transposed, inW, form >= 2);
  if (L.up2) {  // one y parity per launch (a single class: the persistent kernels apply), both x parities as rows
    cy = {axis_classes_up2(inH, true)[L.up2 - 1]};
    cx = axis_classes_up2(inW, false);
  }
  const bool parity_layer = L.transposed || L.up2;
  ConvPlanOut R;
  R.outD = L.transposed ? inD * L.sd : cz[0].npos;
  R.outH = L.transposed ? inH * L.sh : (L.up2 ? 2 * inH : cy[0].npos);
  R.outW = L.transposed ? inW * L.sw : (L.up2 ? 2 * inW : cx[0].npos);
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
  const bool strided[3] = {L.transposed && L.sd == 2, (L.transposed && L.sh == 2) || L.up2, (L.transposed && L.sw == 2) || L.up2};
  const bool dense[3] = {strided[0] && form < 1, strided[1] && form < 1 && !L.up2, strided[2] && (L.up2 || form < 2)};  // parity carried by the output rows
  if (parity_layer) {
    if (mode != CONV_NORMAL) fail(DR_ERR_ARG, "plan_conv: transposed layers use CONV_NORMAL");
    for (int d = 0; d < 3; ++d) if (dense[d]) npar *= 2;
    for (int q = 0; q < npar; ++q) {
      int bits = 0, rem = q;
      for (int d = 2; d >= 0; --d) if (dense[d]) { bits |= (rem & 1) << (2 - d); rem >>= 1; }
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
  const int SZ = cz[0].s, SX = cx[0].s, PZ = cz[0].p, PX = cx[0].p;
  int SY = cy[0].s, PY = cy[0].p;  // (a k_conv_w plan re-describes the y axis once it is chosen)
  const int nPD = cz[0].npos, nPW = cx[0].npos;
  int nPH = cy[0].npos;
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
  int CI = 0, PT = 0, CT = 0, TZ = 0, TY = 0, TXT = 0, TZI = 0, TYI = 0, TXI = 0, ASYNC = 0;
  struct Cand { double cost; int ci, pt, ct, tz, ty, txt, tzi, tyi, txi, async; };
  std::vector<Cand> cands;
  const bool bf3 = conv_bf3_policy() && !fz && L.Cin % 8 == 0;  // (the Cin = 4 first layer and the fused-skip form stay on the fp32 kernels)
  const int policy = (fz || bf3) ? 0 : conv_async_policy();
  if (policy >= 1 && ncls == 1) {  // k_conv_a: 8 waves, 8*pt position tiles per workgroup, two tile buffers
    for (int ci : {16, 8, 4}) {
      if (in_split && ci != in_split) continue;  // a split input only has 16-channel passes
      if (L.Cin % ci || (ci == 4 && L.Cin != 4)) continue;
      const int npass = L.Cin / ci, tpc = 16 / ci, nu = cdiv(classes[0].ntaps, tpc);
      if (npass > 2 && (npass & 1)) continue;  // weight buffers alternate with the pass parity
      for (int pt : {1, 2, 4})
        for (int tz = 1; tz <= 8 * pt; tz *= 2)
          for (int ty = 1; tz * ty <= 8 * pt; ty *= 2) {
            const int txt = 8 * pt / (tz * ty);
            if ((tz > 1 && tz / 2 >= nPD) || (ty > 1 && ty / 2 >= nPH) || (txt > 1 && (txt / 2) * 16 >= nPW)) continue;  // more than half of the tile outside
            const int tzi = (tz - 1) * SZ + exz, tyi = (ty - 1) * SY + exy, txi = (txt * 16 - 1) * SX + exx;
            if ((size_t)tzi * tyi * txi >= 65536) continue;
            const double tiles = (double)cdiv(nPD, tz) * cdiv(nPH, ty) * cdiv(nPW, txt * 16);
            for (int ct : {2, 1}) {
              if (CTtot % ct || !conv_a_instance_exists(ci, ct, pt)) continue;
              const size_t slots = conv_a_slots(tzi * tyi * txi, ci);
              size_t bytes = 2 * slots * 16 + (size_t)npass * nu * ct * 1024 + (size_t)nu * tpc * 4 + 64;  // all passes' weights resident ...
              if (bytes > kConvMaxLds) bytes = 2 * slots * 16 + (size_t)std::min(npass, 2) * nu * ct * 1024 + (size_t)nu * tpc * 4 + 64;  // ... or two in flight
              if (bytes > kConvMaxLds) continue;
              const double wpc = bytes * 2 <= kConvMaxLds ? 2.0 : 1.0, split = CTtot / ct;
              const double mfma_unit = nu * 4.0 * ct * pt * 32.0 * 2.0;  // cycles per SIMD: two waves of the workgroup share it
              const double stage_unit = slots / 4.0;                    // ~64 B/clk/CU from L2
              const double unit = wpc * std::max(mfma_unit, stage_unit) + 700.0;
              // x 1.1: the two families' cost models are not calibrated against each other; an untuned shape only
              // moves to the persistent kernel when its model says so with some margin
              const double cost = 1.1 * std::ceil(tiles * split / (256.0 * wpc)) * npass * unit;
              cands.push_back({cost, ci, pt, ct, tz, ty, txt, tzi, tyi, txi, 1});
            }
          }
    }
  }
  // k_conv_m: the stride-1 3x3 / 3x3x3 layers on the marching producer/consumer kernel (conv_march.h)
  // (a fused FeatureNet skip runs on it in exactly one form: 8-channel source, 32 -> 8 XPAIR 3x3 layer, producers compute the tile)
  const bool march_fz_ok = !fz || (fz->cin == 8 && L.Cin == 32 && L.kd == 1 && mode == CONV_XPAIR && !hook_env("DR_FZ_NO_MARCH"));
  const int march_policy = (march_fz_ok && !bf3) ? conv_march_policy() : 0;
  const bool march_ok = march_policy >= 1 && ncls == 1 && !L.transposed && !L.up2 && mode != CONV_X8 && L.kh == 3 && L.kw == 3 && (L.kd == 1 || L.kd == 3) &&
                        (int)cz[0].t.size() == L.kd &&  // (a depth axis with pruned taps -- extent 1 or 2 -- stays on the tile kernels)
                        SZ == 1 && SY == 1 && L.sw == 1;
  const int march_ntp = march_ok ? 3 * (int)cx[0].t.size() : 0;  // taps per input plane
  if (march_ok) {
    for (int ci : {16, 8}) {
      if (in_split && ci != in_split) continue;  // a split input only has 16-channel passes
      if (ci == 8 && L.Cin != 8) continue;
      for (int ncw : {8, 12})
      for (int pt : {1, 2, 4})
        for (int ty = 1; ty <= ncw * pt; ++ty) {
          if ((ncw * pt) % ty) continue;
          const int txt = ncw * pt / ty;
          if ((ty > 1 && ty / 2 >= nPH) || (txt > 1 && (txt / 2) * 16 >= nPW)) continue;
          for (int ct : {2, 1}) {
            const MarchShape ms = march_shape(L.kd, march_ntp, L.Cin, ci, ct, pt, ty, txt, SX, exy, exx, nPD, nPH, nPW, CTtot);
            if (!ms.ok || (fz && (ci != 16 || ct != 1 || ncw != 8))) continue;
            const double spw = std::ceil((double)ms.steps / ms.grid);
            const double unit = ms.ns * ms.nup * 4.0 * ct * pt * 32.0 * (ncw / 4) * (ncw == 12 ? 0.94 : 1.0);  // MFMA cycles of a step per SIMD (ncw / 4 consumer waves each; three hide each other's gaps better)
            const double startup = (double)ms.lds_bytes / 16.0 + 3000.0;
            double cost = ms.npo * (spw * unit * 1.02 + startup);
            if (fz) cost *= 1.5;  // measured (r3): with a 3-slot ring and two channel passes per tile the fused-skip producers are the bottleneck (0.24 ms against k_conv's 0.214 at 480 x 640)
            else if (L.kd == 1) cost *= 1.6;  // 2-D layers: one step per tile, the whole halo per step -- measured 1.1-1.25x k_conv_a's time where the model says 0.6x (profiles/r03_experiments.txt, 8)
            if (march_policy >= 2) cost *= 1e-3;
            cands.push_back({cost, ci, pt, ct, 1, ty, txt, L.kd, ms.tyi, ms.txi, 2});
          }
        }
    }
  }
  // the Winograd form of the marching kernel (async = 5): 3-D layers, even output height; ty counts row pairs
  const int wino_policy_m = (!fz && !bf3) ? conv_wino_policy() : 0;
  if (march_ok && wino_policy_m >= 1 && L.kd == 3 && (R.outH & 1) == 0 && R.outH >= 2) {
    const int nPHw = R.outH / 2;
    for (int ci : {16, 8}) {
      if (in_split && ci != in_split) continue;  // a split input only has 16-channel passes
      if (ci == 8 && L.Cin != 8) continue;
      for (int ncw : {8})
        for (int ty = 1; ty <= ncw; ++ty) {
          if (ncw % ty) continue;
          const int txt = ncw / ty;
          if ((ty > 1 && ty / 2 >= nPHw) || (txt > 1 && (txt / 2) * 16 >= nPW)) continue;
          const MarchShape ms = march_shape(L.kd, march_ntp, L.Cin, ci, 1, 1, ty, txt, SX, exy, exx, nPD, nPHw, nPW, CTtot, false, true);
          if (!ms.ok) continue;
          const double spw = std::ceil((double)ms.steps / ms.grid);
          const double unit = ms.ns * (ms.nup / 3) * 16.0 * 32.0 * (ncw / 4) * (ncw == 12 ? 0.94 : 1.0);  // MFMA cycles of a step per SIMD: 16 per chunk of x taps and section
          const double startup = (double)ms.lds_bytes / 16.0 + 3000.0;
          double cost = ms.npo * (spw * unit * 1.1 + startup);
          if (wino_policy_m >= 2) cost *= 1e-3 * 0.5;  // (preferred: ahead of k_conv_w's candidates as well)
          cands.push_back({cost, ci, 1, 1, 1, ty, txt, L.kd, ms.tyi, ms.txi, 5});
        }
    }
  }
  // the row march: 2-D 3x3 stride-1 layers on the same kernel, marching down the rows of each image (async = 3)
  const bool rowmarch_ok = !fz && !bf3 && conv_rowmarch_policy() >= 1 && ncls == 1 && !L.transposed && !L.up2 && mode != CONV_X8 && L.kd == 1 && L.kh == 3 && L.kw == 3 &&
                           SZ == 1 && SY == 1 && L.sw == 1 && !add;
  const int row_ntp = rowmarch_ok ? (int)cx[0].t.size() : 0;  // x taps of one row
  if (rowmarch_ok) {
    for (int ci : {16, 8}) {
      if (in_split && ci != in_split) continue;  // a split input only has 16-channel passes
      if (ci == 8 && L.Cin != 8) continue;
      for (int ncw : {8, 10})
        for (int pt : {1, 2, 4}) {
          const int txt = ncw * pt;
          if (txt > 1 && (txt / 2) * 16 >= nPW) continue;
          for (int ct : {2, 1}) {
            const MarchShape ms = march_shape(L.kd, row_ntp, L.Cin, ci, ct, pt, 1, txt, SX, exy, exx, nPD, nPH, nPW, CTtot, true);
            if (!ms.ok) continue;
            const double spw = std::ceil((double)ms.steps / ms.grid);
            const double waste = (double)cdiv(nPW, txt * 16) * txt * 16 / nPW;  // strips hanging over the end of the row
            const double unit = ms.ns * ms.nup * 4.0 * ct * pt * 32.0 * (ncw / 4.0) + 300.0;  // MFMA cycles of a step per SIMD + the per-step bookkeeping
            const double startup = (double)ms.lds_bytes / 16.0 + 3000.0;
            double cost = (spw * unit * 1.02 + startup) * 1.5;  // measured 1.05-1.15x k_conv_a's time at 6-13 steps per workgroup: the model has no term for the ring fill
            (void)waste;
            if (conv_rowmarch_policy() >= 2) cost *= 1e-3;
            cands.push_back({cost, ci, pt, ct, 1, 1, txt, 1, 1, ms.txi, 3});
          }
        }
    }
  }
  // k_conv_w: k_conv's launch form with the y axis in Winograd F(2,3) form (conv_wino.h).  A position tile is 16 x by one row PAIR:
  // to the tile geometry the layer has stride 2 and a 4-row kernel in y; a K chunk is 4 points x 4 MFMAs per (row tile, position tile).
  const int wino_policy = (!fz && !bf3) ? conv_wino_policy() : 0;
  const bool wino_ok = wino_policy >= 1 && ncls == 1 && !L.transposed && !L.up2 && mode != CONV_X8 && L.kh == 3 && L.sh == 1 && (R.outH & 1) == 0 && R.outH >= 2;
  if (wino_ok) {
    static const int cand8[][3] = {{1, 1, 8}, {1, 2, 4}, {1, 4, 2}, {1, 8, 1}, {2, 1, 4}, {2, 2, 2}, {2, 4, 1}, {4, 1, 2}, {4, 2, 1}, {8, 1, 1}};
    const int nPHw = R.outH / 2, ntw = (int)(cz[0].t.size() * cx[0].t.size());
    for (int ci : {16, 8}) {
      if (in_split && ci != in_split) continue;  // a split input only has 16-channel passes
      if (L.Cin % ci || (ci == 8 && L.Cin != 8)) continue;
      const int npass = L.Cin / ci, tpc = 16 / ci, nr = cdiv(ntw, tpc);
      for (int pt : {2, 1}) {
        const int (*cand)[3] = pt == 2 ? cand8 : cand4;
        const int ncand = pt == 2 ? 10 : 6;
        for (int k = 0; k < ncand; ++k) {
          const int *c = cand[k];
          if ((c[0] > 1 && c[0] / 2 >= nPD) || (c[1] > 1 && c[1] / 2 >= nPHw) || (c[2] > 1 && (c[2] / 2) * 16 >= nPW)) continue;  // more than half of the tile outside
          const int tzi = (c[0] - 1) * SZ + exz, tyi = (c[1] - 1) * 2 + 4, txi = (c[2] * 16 - 1) * SX + exx;
          if ((size_t)tzi * tyi * txi >= 65536) continue;
          const double tiles = (double)cdiv(nPD, c[0]) * cdiv(nPHw, c[1]) * cdiv(nPW, c[2] * 16);
          for (int ct : {2, 1}) {
            if (CTtot % ct || !conv_w_instance_exists(ci, ct, pt)) continue;
            const size_t bytes = (size_t)tzi * tyi * txi * (ci + 4) * 4 + (size_t)4 * nr * ct * 1024 + (size_t)4 * nr * tpc * 4 + 64;
            if (bytes > kConvMaxLds) continue;
            const int split = CTtot / ct;
            const double waves_simd = ct * pt == 1 ? 4.0 : (ct * pt == 2 ? 3.0 : 2.0);  // register budget of the instance
            const double wg_per_cu = std::max(1.0, std::min({(double)(kConvMaxLds / bytes), waves_simd}));
            const double stage = npass * (((double)tzi * tyi * txi * (ci / 4) + 4.0 * nr * ct * 64.0) / 256.0 * 60.0 + 900.0);
            const double chunk_mfma = 16.0 * ct * pt * 32.0;
            const double n_wg = tiles * split;
            const double mfma_total = n_wg * npass * nr * chunk_mfma, stage_total = n_wg * stage;
            const double lat_wg = npass * nr * chunk_mfma + stage;
            const double waves = std::ceil(n_wg / (256.0 * wg_per_cu));
            const double thr = (mfma_total + stage_total) / 256.0 / (wg_per_cu >= 2 ? 0.8 : 0.5);
            // measured (r4, profiles/r04_winograd.txt): 14-17 % faster than the best direct plan on every 2-D 3x3 layer of both tuned shapes, 5-7 % on the
            // 16- to 64-channel 3-D layers, slower than the marching kernel on the conv0 layers.  The model below overstates the gain (it has no term
            // for the smaller tiles' halo), so an untuned shape moves here with a margin, and a 3-D layer only when a tuned row says so.
            double cost = std::max(thr, waves * lat_wg) * (L.kd == 1 ? 1.25 : 2.5);
            if (wino_policy >= 2) cost *= 1e-3;
            cands.push_back({cost, ci, pt, ct, c[0], c[1], c[2], tzi, tyi, txi, 4});
          }
        }
      }
    }
  }
  const bool sync_too = policy != 1 || cands.empty();
  for (int ci : {16, 8, 4}) {
    if (in_split && ci != in_split) continue;  // a split input only has 16-channel passes
    if (!sync_too) break;
    if (L.Cin % ci || (ci == 4 && L.Cin != 4)) continue;
    if (fz && ci != 16) continue;  // fused-skip instances exist for 16-channel passes, one row tile
    if (bf3 && ci == 4) continue;
    const int npass = L.Cin / ci, tpc = (bf3 ? 32 : 16) / ci, wkb = bf3 ? 2048 : 1024;  // taps and weight bytes per K chunk (and 16 output rows)
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
          if (CTtot % ct || !conv_instance_exists(ci, ct) || (fz && ct != 1)) continue;
          const size_t bytes = (size_t)tzi * tyi * txi * (ci + 4) * 4 + (size_t)nu_max * ct * wkb + (size_t)nu_max * tpc * 4 + 64;
          if (bytes > kConvMaxLds) continue;
          const int split = CTtot / ct;
          // cost model (cycles): MFMA issue, staging, and a latency floor per chunk; see DESIGN.md
          const double wg_per_cu = std::max(1.0, std::min({(double)(kConvMaxLds / bytes), 8.0, (ct == 4 && pt == 4) ? 5.0 : 8.0}));
          const double stage = npass * (((double)tzi * tyi * txi * (ci / 4) + (chunks / ncls) * ct * 64.0) / 256.0 * 60.0 + 900.0);
          const double chunk_mfma = bf3 ? 3.0 * ct * pt * 16.0 : 4.0 * ct * pt * 32.0;  // (bf3: three 16-cycle MFMAs per 32-wide chunk)
          const double n_wg = tiles * split;  // per class
          const double mfma_total = n_wg * npass * chunks * chunk_mfma, stage_total = n_wg * ncls * stage;
          const double lat_wg = npass * (chunks / ncls) * std::max(chunk_mfma, 160.0) + stage;
          const double waves = std::ceil(n_wg * ncls / (256.0 * wg_per_cu));
          const double thr = (mfma_total + stage_total) / 256.0 / (wg_per_cu >= 2 ? 0.8 : 0.5);
          const double cost = std::max(thr, waves * lat_wg);
          cands.push_back({cost, ci, pt, ct, c[0], c[1], c[2], tzi, tyi, txi, 0});
        }
      }
    }
  }
  if (in_split) {
    if (in_split != 16 || fz || bf3 || L.Cin % 16) fail(DR_ERR_ARG, "plan_conv: a split input has 16-channel sub-tensors and plain fp32 staging");
  }
  if (!cands.empty()) {
    std::stable_sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.cost < b.cost; });
    if (!getenv("DR_CONV_NO_TUNED")) {  // a measured plan for exactly this layer swaps places with the cost model's first choice -- for EVERY rank, so that the
                                        // ranks stay a permutation of the candidates (the autotuner used to never time the model's own first choice of a tuned layer)
      for (const ConvTuned &t : kConvTuned) {
        if (t.Cin != L.Cin || t.Cout != L.Cout || t.kd != L.kd || t.kh != L.kh || t.kw != L.kw || t.sd != L.sd || t.sh != L.sh || t.sw != L.sw ||
            t.transposed != (L.transposed ? 1 : (L.up2 ? 1 + L.up2 : 0)) || t.mode != (int)mode + (in_split ? 16 : 0) || t.inD != inD || t.inH != inH || t.inW != inW) continue;
        for (size_t i = 0; i < cands.size(); ++i) {
          const Cand &k = cands[i];
          if (k.ci == t.ci && k.ct == t.ct && k.pt == t.pt && k.tz == t.tz && k.ty == t.ty && k.txt == t.txt && k.async == t.async_) { std::swap(cands[0], cands[i]); break; }
        }
        break;
      }
    }
    const Cand &k = cands[std::min<size_t>(rank < 0 ? 0 : rank, cands.size() - 1)];
    best = k.cost; CI = k.ci; PT = k.pt; CT = k.ct; TZ = k.tz; TY = k.ty; TXT = k.txt; TZI = k.tzi; TYI = k.tyi; TXI = k.txi; ASYNC = k.async;
    R.ncand = (int)cands.size();
  }
  if (!CI) fail(DR_ERR_ARG, "plan_conv: no kernel instance / tile shape for Cin=%d Cout=%d", L.Cin, L.Cout);
  const int npass = L.Cin / CI, TPC = (bf3 ? 32 : 16) / CI, CIS = CI + 4;
  const bool wino = ASYNC == 4 || ASYNC == 5, wino_raw = ASYNC == 5;  // 5: the marching kernel derives the transformed weights itself (raw rows g0, g1 / 2, g2)
  if (wino) {  // the y axis as k_conv_w sees it: positions are row pairs, the tap table carries row 0 only (the kernel adds rows 1..3 itself)
    DimTaps &Y = cy[0];
    Y.t = {0}; Y.off = {0}; Y.s = 2; Y.p = 1; Y.om = 2; Y.oo = 0; Y.npos = R.outH / 2;
    SY = 2; PY = 1; nPH = Y.npos;
    classes[0].ntaps = (int)(cz[0].t.size() * cx[0].t.size());
  }

  // per-row epilogue affine
  std::vector<float> sc(rows, 1.f), bi(rows, 0.f);
  for (int r = 0; r < rows_valid; ++r) {
    const int c = parity_layer ? r % L.Cout : (mode == CONV_NORMAL ? r : (mode == CONV_XPAIR ? (r & 7) : 0));
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
    const int NR = cdiv(ntaps, TPC), NU = wino_raw ? 3 * NR : (wino ? 4 * NR : NR);  // k_conv_w: four weight chunks (one per Winograd point) per chunk of taps; marching form: three (the raw kernel rows)
    cls[ic].NU = NU; cls[ic].tap_base = (int)tapoff.size(); cls[ic].w_base = (int)(pk.size() / 4);
    cls[ic].ooz = Z.oo; cls[ic].ooy = Y.oo; cls[ic].oox = X.oo;
    std::vector<int> tz(ntaps), ty(ntaps), tx(ntaps);
    const size_t t0 = tapoff.size();
    tapoff.resize(t0 + (size_t)NR * TPC, 0);
    {
      int n = 0;
      for (int iz = 0; iz < ntz; ++iz) for (int iy = 0; iy < nty; ++iy) for (int ix = 0; ix < ntx; ++ix, ++n) {
        tapoff[t0 + n] = (Z.off[iz] * TYI + Y.off[iy]) * TXI + X.off[ix];
        tz[n] = Z.t[iz]; ty[n] = Y.t[iy]; tx[n] = X.t[ix];
      }
    }
    const size_t w0 = pk.size();
    auto weight_of = [&](int tap, int cin, int row, int ky = -1) -> float {  // the weight that multiplies input channel `cin` of tap `tap` for output row `row` (ky >= 0: that kernel row instead of the tap's)
      float v = 0.f;
      if (tap < ntaps && row < rows_valid) {
        if (L.up2) {  // sum of the kernel entries that fall on this (parity, input offset) pair, per axis
          const int q = row / L.Cout, co = row % L.Cout, bits = (par_map >> (3 * q)) & 7;
          int ky[2], kx[2];
          const int ny = up2_kernel_set(Y.par, ty[tap], ky), nx = up2_kernel_set(bits & 1, tx[tap], kx);
          double acc = 0;
          for (int iy = 0; iy < ny; ++iy) for (int ix = 0; ix < nx; ++ix) acc += weight_at(co, cin, 0, ky[iy], kx[ix]);
          v = (float)acc;
        } else if (L.transposed) {
          const int q = row / L.Cout, co = row % L.Cout, bits = (par_map >> (3 * q)) & 7;
          const int offs[3] = {tz[tap], ty[tap], tx[tap]};  // for transposed layers DimTaps::t carries the input offset
          int kk[3];
          bool ok = true;
          const int cpar[3] = {Z.par, Y.par, X.par};
          for (int d = 0; d < 3; ++d) {
            if (strided[d]) kk[d] = parity_kernel_index(dense[d] ? (bits >> (2 - d)) & 1 : cpar[d], offs[d]);
            else kk[d] = offs[d];  // stride-1 axis: DimTaps::t is the kernel index already (k == 1 or the 3-tap flip)
            ok = ok && kk[d] >= 0;
          }
          if (ok) v = weight_at(co, cin, kk[0], kk[1], kk[2]);
        } else if (mode == CONV_NORMAL) v = weight_at(row, cin, tz[tap], ky >= 0 ? ky : ty[tap], tx[tap]);
        else {
          const int shift = mode == CONV_XPAIR ? (row >> 3) : row, co = mode == CONV_XPAIR ? (row & 7) : 0;
          const int kx = tx[tap] - shift;
          if (kx >= 0 && kx < L.kw) v = weight_at(co, cin, tz[tap], ky >= 0 ? ky : ty[tap], kx);
        }
      }
      return v;
    };
    pk.resize(w0 + (size_t)npass * NU * CTtot * 64 * 4 * (bf3 ? 2 : 1), 0.f);
    if (bf3) {  // [pass][chunk][row tile][hi | lo][lane] x 8 bf16: lane (i = l & 15, g = l >> 4) holds K = 32 u + 8 g .. + 7 of row 16 ct + i
      unsigned short *pb = reinterpret_cast<unsigned short *>(pk.data() + w0);
      for (int p = 0; p < npass; ++p) for (int u = 0; u < NU; ++u) for (int ct = 0; ct < CTtot; ++ct)
        for (int l = 0; l < 64; ++l) for (int s = 0; s < 8; ++s) {
          const int g = l >> 4, i = l & 15, k32 = 8 * g + s;
          const float v = weight_of(u * TPC + k32 / CI, p * CI + k32 % CI, ct * 16 + i);
          const unsigned short hi = bf16_rne(v), lo = bf16_rne(v - bf16_value(hi));
          const size_t frag = (((size_t)p * NU + u) * CTtot + ct) * 2;
          pb[((frag + 0) * 64 + l) * 8 + s] = hi;
          pb[((frag + 1) * 64 + l) * 8 + s] = lo;
        }
    } else if (wino_raw) {  // [pass][chunk of taps][kernel row k][row tile][lane]: g0, g1 / 2, g2 (march_consumer_w forms u1, u2 from them)
      for (int p = 0; p < npass; ++p) for (int u = 0; u < NR; ++u) for (int k = 0; k < 3; ++k) for (int ct = 0; ct < CTtot; ++ct)
        for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) {
          const int g = l >> 4, i = l & 15, k16 = 4 * g + s;
          pk[w0 + ((((size_t)p * NU + u * 3 + k) * CTtot + ct) * 64 + l) * 4 + s] = (k == 1 ? 0.5f : 1.f) * weight_of(u * TPC + k16 / CI, p * CI + k16 % CI, ct * 16 + i, k);
        }
    } else if (wino) {  // [pass][chunk of taps][point][row tile][lane]: u_p = sum_ky G[p][ky] w[ky], formed in double, rounded once
      for (int p = 0; p < npass; ++p) for (int u = 0; u < NR; ++u) for (int pp = 0; pp < 4; ++pp) for (int ct = 0; ct < CTtot; ++ct)
        for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) {
          const int g = l >> 4, i = l & 15, k16 = 4 * g + s;
          double v = 0;
          for (int ky = 0; ky < 3; ++ky) v += conv_wino_g(pp, ky) * (double)weight_of(u * TPC + k16 / CI, p * CI + k16 % CI, ct * 16 + i, ky);
          pk[w0 + ((((size_t)p * NU + u * 4 + pp) * CTtot + ct) * 64 + l) * 4 + s] = (float)v;
        }
    } else
    for (int p = 0; p < npass; ++p) for (int u = 0; u < NU; ++u) for (int ct = 0; ct < CTtot; ++ct)
      for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) {
        const int g = l >> 4, i = l & 15, k16 = 4 * g + s;
        pk[w0 + ((((size_t)p * NU + u) * CTtot + ct) * 64 + l) * 4 + s] = weight_of(u * TPC + k16 / CI, p * CI + k16 % CI, ct * 16 + i);
      }
    if (L.up2) flops += ic == 0 ? 2.0 * nPD * nPH * nPW * 16.0 * L.Cin * L.Cout : 0.0;  // 4 output pixels x (2 x 2 input pixels) per input position
    else if (L.transposed) flops += ic == 0 ? 2.0 * nPD * nPH * nPW * (double)L.kd * L.kh * L.kw * L.Cin * L.Cout : 0.0;
    else flops += 2.0 * nPD * (wino ? 2 * nPH : nPH) * nPW * (mode == CONV_NORMAL ? 1 : shifts) * (double)ntz * (wino ? 3 : nty) * (mode == CONV_NORMAL ? ntx : L.kw) * L.Cin * L.Cout;  // (algorithmic: the direct form's)
  }

  ConvLaunch cl{};
  ConvArgs &a = cl.args;
  a.in = in; a.out = out; a.wpk = reinterpret_cast<const float4 *>(arena.upload(pk));
  a.scale = arena.upload(sc); a.bias = arena.upload(bi); a.add = add; a.tapoff = arena.upload(tapoff);
  a.cls = arena.upload(cls);
  a.inD = inD; a.inH = inH; a.inW = inW; a.inC = in_split ? in_split : inC;
  a.pass_stride = in_split ? (int)((size_t)inD * inH * inW * in_split) : CI;
  if (in_split && (size_t)inD * inH * inW * in_split >= (1ull << 31)) fail(DR_ERR_ARG, "plan_conv: split input too large");
  a.outD = R.outD; a.outH = R.outH + 2 * L.out_pad; a.outW = outWv; a.outC = outCv;
  if (L.out_pad) {  // bordered output: only the strides change (the x border is a whole number of XPAIR / X8 output groups or the mode is refused)
    if ((2 * L.out_pad) % shifts) fail(DR_ERR_ARG, "plan_conv: an output border of %d pixels does not fit the %d-wide output groups", L.out_pad, shifts);
    a.outW = outWv + 2 * L.out_pad / shifts;
  }
  a.nPD = nPD; a.nPH = nPH; a.nPW = nPW;
  a.sz = SZ; a.sy = SY; a.sx = SX; a.pz = PZ; a.py = PY; a.px = PX;
  a.omz = cz[0].om; a.omy = cy[0].om; a.omx = cx[0].om;
  a.par_rows = parity_layer ? L.Cout : 0; a.par_map = par_map;
  a.TZ = TZ; a.TY = TY; a.TXT = TXT; a.TZI = TZI; a.TYI = TYI; a.TXI = TXI;
  a.magicX = TXI == 1 ? 0u : (unsigned)((0x100000000ull + TXI - 1) / TXI);
  a.magicY = TYI == 1 ? 0u : (unsigned)((0x100000000ull + TYI - 1) / TYI);
  if ((size_t)TZI * TYI * TXI >= 65536) fail(DR_ERR_ARG, "plan_conv: halo tile too large");
  a.npass = npass; a.ctTot = CTtot; a.rows_valid = rows_valid; a.relu = L.relu ? 1 : 0;
  a.add_mode = add ? add_mode : 0;
  a.addH = R.outH / 2; a.addW = (mode == CONV_NORMAL ? R.outW : outWv) / 2;
  a.tilesD = cdiv(nPD, TZ); a.tilesH = cdiv(nPH, TY); a.tilesW = cdiv(nPW, TXT * 16);
  cl.ci = CI; cl.ct = CT; cl.pt = PT;
  a.fz_x = a.fz_w = a.fz_b = a.fz_coarse = nullptr;
  if (fz) {
    if (fz->cin != 8 || L.transposed || SZ != 1 || SY != 1) fail(DR_ERR_ARG, "plan_conv: fused skip needs an 8-channel source and a stride-1 layer");
    a.fz_x = fz->x; a.fz_w = fz->w; a.fz_b = fz->b; a.fz_coarse = fz->coarse; cl.fz = fz->cin;
  }
  cl.grid = dim3(8 * cdiv(a.tilesD * a.tilesH * a.tilesW, 8), ncls, CTtot / CT);
  int nu_max = 0;
  for (auto &c : cls) nu_max = std::max(nu_max, c.NU);
  a.nuMax = nu_max;
  cl.lds_bytes = (size_t)TZI * TYI * TXI * CIS * 4 + (size_t)nu_max * CT * (bf3 ? 2048 : 1024) + (size_t)nu_max * TPC * 4 + 64;
  cl.bf3 = bf3 ? 1 : 0;
  a.zero16 = nullptr; a.a_slots = 0; a.a_wbufs = 1; a.class_loop = 0;
  // k_conv's class loop (see the kernel): single-pass layers with several parity classes, both weight buffers and every class's tap table next to the tile
  if (ASYNC == 0 && !bf3 && !fz && ncls > 1 && npass == 1 && conv_c_instance_exists(CI, CT, PT) && !hook_env("DR_CONV_NO_CLASS_LOOP")) {
    const size_t with_loop = cl.lds_bytes + (size_t)nu_max * CT * 1024 + (size_t)(ncls - 1) * nu_max * TPC * 4;
    if (with_loop <= kConvMaxLds) {
      a.class_loop = ncls;
      cl.lds_bytes = with_loop;
      cl.grid.y = 1;
    }
  }
  // k_conv with pipelined passes (see the kernel): small multi-pass layers on the one-position-tile instances whose tile is at most 6 loads per lane
  if (ASYNC == 0 && !bf3 && !fz && PT == 1 && CT <= 2 && npass >= 2 && (size_t)TZI * TYI * TXI * (CI / 4) <= 6 * kConvThreads && !hook_env("DR_CONV_NO_PIPE") &&
      cl.lds_bytes + (size_t)nu_max * CT * 1024 <= kConvMaxLds) {
    a.a_wbufs = 2;
    cl.lds_bytes += (size_t)nu_max * CT * 1024;
  }
  if (ASYNC == 4) cl.async = 4;  // (k_conv's LDS layout and grid: the tile, nuMax weight chunks, the tap table)
  if (ASYNC == 1) {
    cl.async = 1;
    a.zero16 = arena.upload(std::vector<float>(4, 0.f));
    a.a_slots = (int)conv_a_slots(TZI * TYI * TXI, CI);
    a.a_wbufs = npass;
    cl.lds_bytes = 2 * (size_t)a.a_slots * 16 + (size_t)a.a_wbufs * nu_max * CT * 1024 + (size_t)nu_max * TPC * 4 + 64;
    if (cl.lds_bytes > kConvMaxLds) {  // the passes' weights do not all fit next to the two tile buffers: two buffers, re-fetched per pass
      a.a_wbufs = std::min(npass, 2);
      cl.lds_bytes = 2 * (size_t)a.a_slots * 16 + (size_t)a.a_wbufs * nu_max * CT * 1024 + (size_t)nu_max * TPC * 4 + 64;
    }
    const int ntiles = a.tilesD * a.tilesH * a.tilesW, split = CTtot / CT;
    const int wpc = cl.lds_bytes * 2 <= kConvMaxLds ? 2 : 1;
    const int want = std::max(1, std::min(ntiles, 256 * wpc / split));
    cl.grid = dim3(8 * cdiv(want, 8), 1, split);
  }
  if (ASYNC == 2 || ASYNC == 3 || ASYNC == 5) {
    const bool rm = ASYNC == 3;
    const MarchShape ms = rm ? march_shape(L.kd, row_ntp, L.Cin, CI, CT, PT, 1, TXT, SX, exy, exx, nPD, nPH, nPW, CTtot, true)
                             : march_shape(L.kd, march_ntp, L.Cin, CI, CT, PT, TY, TXT, SX, exy, exx, nPD, nPH, nPW, CTtot, false, wino_raw);
    if (!ms.ok) fail(DR_ERR_ARG, "plan_conv: inconsistent k_conv_m plan");
    cl.async = 2; cl.nup = ms.nup; cl.ncw = ms.ncw;
    a.zero16 = arena.upload(std::vector<float>(4, 0.f));
    a.a_slots = 0; a.a_wbufs = 0;
    MarchArgs &m = cl.march;
    const int kz = rm ? 3 : L.kd;
    std::vector<int> tap2d((size_t)ms.nup * TPC, 0);
    {
      const DimTaps &Y = *classes[0].y, &X = *classes[0].x;
      int n = 0;
      if (rm) for (size_t ix = 0; ix < X.t.size(); ++ix, ++n) tap2d[n] = X.off[ix];  // the y taps are the sections of the step
      else for (size_t iy = 0; iy < Y.t.size(); ++iy) for (size_t ix = 0; ix < X.t.size(); ++ix, ++n) tap2d[n] = Y.off[iy] * TXI + X.off[ix];
    }
    m.tap2d = arena.upload(tap2d);
    m.geo.KZ = kz; m.geo.NPI = ms.npi; m.geo.Dc = rm ? nPH : (L.kd == 3 ? nPD : 1);
    m.NPO = ms.npo;
    m.colsH = rm ? 1 : cdiv(nPH, TY); m.colsW = cdiv(nPW, TXT * 16);
    m.ncols = (L.kd == 3 ? 1 : nPD) * m.colsH * m.colsW;
    m.R = ms.r; m.PS = ms.ps; m.NP = ms.np; m.nit = ms.ps / 128;
    m.wsec = ms.nup * CT * 64; m.NU = kz * ms.nup;
    m.steps = (int)ms.steps;
    m.ncw = ms.ncw;
    m.rm = rm ? 1 : 0;
    m.wino = wino_raw ? 1 : 0;
    const long long iplane = (long long)a.inH * a.inW * a.inC, oplane = (long long)a.outH * a.outW * a.outC;
    if (iplane * std::max(1, a.inD) >= (1ll << 31) || oplane * std::max(1, a.outD) >= (1ll << 31)) fail(DR_ERR_ARG, "plan_conv: tensor too large for k_conv_m's 32-bit strides");
    m.i_sv = L.kd == 3 ? 0 : (int)iplane; m.i_sz = rm ? a.inW * a.inC : (L.kd == 3 ? (int)iplane : 0); m.i_sy = rm ? 0 : a.inW * a.inC;
    m.o_sv = L.kd == 3 ? 0 : (int)oplane; m.o_sz = rm ? a.outW * a.outC : (L.kd == 3 ? (int)oplane : 0); m.o_sy = rm ? 0 : a.outW * a.outC;
    m.inHp = rm ? 1 : a.inH;
    m.err = arena.err_flag;
    m.depth = rm ? 3 : 2;  // loads each producer wave keeps in flight (rows are small and steps short: one more)
    if (const char *e = hook_env("DR_MARCH_PDEPTH")) m.depth = std::max(1, std::min(4, atoi(e)));  // A/B hook
    cl.lds_bytes = ms.lds_bytes;
    cl.grid = dim3(ms.grid, 1, CTtot / CT);
  }
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
template <int CI, int CT, int PT, int FZ = 0>
inline void launch_conv_inst(const ConvLaunch &c, hipStream_t st) {
  static std::atomic<unsigned long long> done{0};  // bit d: set on device d
  conv_allow_big_lds(reinterpret_cast<const void *>(&k_conv<CI, CT, PT, FZ>), done, c.lds_bytes);
  hipLaunchKernelGGL((k_conv<CI, CT, PT, FZ>), c.grid, dim3(kConvThreads), c.lds_bytes, st, c.args);
}
#ifdef DR_PARITY_HOOKS
template <int CI, int CT, int PT>
inline void launch_conv_b_inst(const ConvLaunch &c, hipStream_t st) {
  static std::atomic<unsigned long long> done{0};
  conv_allow_big_lds(reinterpret_cast<const void *>(&k_conv_b<CI, CT, PT>), done, c.lds_bytes);
  hipLaunchKernelGGL((k_conv_b<CI, CT, PT>), c.grid, dim3(kConvThreads), c.lds_bytes, st, c.args);
}
#endif
template <int CI, int CT, int PT>
inline void launch_conv_c_inst(const ConvLaunch &c, hipStream_t st) {
  static std::atomic<unsigned long long> done{0};
  conv_allow_big_lds(reinterpret_cast<const void *>(&k_conv_c<CI, CT, PT>), done, c.lds_bytes);
  hipLaunchKernelGGL((k_conv_c<CI, CT, PT>), c.grid, dim3(kConvThreads), c.lds_bytes, st, c.args);
}
template <int CI, int CT, int PT>
inline void launch_conv_a_inst(const ConvLaunch &c, hipStream_t st) {
  static std::atomic<unsigned long long> done{0};
  conv_allow_big_lds(reinterpret_cast<const void *>(&k_conv_a<CI, CT, PT>), done, c.lds_bytes);
  hipLaunchKernelGGL((k_conv_a<CI, CT, PT>), c.grid, dim3(kConvAThreads), c.lds_bytes, st, c.args);
}
template <int CI, int CT, int PT>
inline void launch_conv_w_inst(const ConvLaunch &c, hipStream_t st) {
  static std::atomic<unsigned long long> done{0};
  conv_allow_big_lds(reinterpret_cast<const void *>(&k_conv_w<CI, CT, PT>), done, c.lds_bytes);
  hipLaunchKernelGGL((k_conv_w<CI, CT, PT>), c.grid, dim3(kConvThreads), c.lds_bytes, st, c.args);
}
template <int CI, int NUP, int CT, int PT, int FZ = 0, int NCW = 8, int W = 0>
inline void launch_conv_m_inst(const ConvLaunch &c, hipStream_t st) {
  static std::atomic<unsigned long long> done{0};
  conv_allow_big_lds(reinterpret_cast<const void *>(&k_conv_m<CI, NUP, CT, PT, FZ, NCW, W>), done, c.lds_bytes);
  hipLaunchKernelGGL((k_conv_m<CI, NUP, CT, PT, FZ, NCW, W>), c.grid, dim3(64 * (NCW + (FZ ? kMarchFzProducers : kMarchProducers))), c.lds_bytes, st, c.args, c.march);
}
inline void launch_conv(const ConvLaunch &c, hipStream_t st) {
#ifndef DR_PARITY_HOOKS
  if (c.bf3) fail(DR_ERR_UNSUPPORTED, "launch_conv: the bf16 x 3 kernel is built with -DDR_PARITY_HOOKS only");
#else
  if (c.bf3) {
#define DR_CONV_B_CASE(CI_, CT_)                                                \
  if (c.ci == CI_ && c.ct == CT_ && !c.async && !c.fz) {                        \
    if (c.pt == 4) launch_conv_b_inst<CI_, CT_, 4>(c, st);                      \
    else launch_conv_b_inst<CI_, CT_, 1>(c, st);                                \
    return;                                                                     \
  }
    DR_CONV_B_CASE(8, 1)
    DR_CONV_B_CASE(8, 2)
    DR_CONV_B_CASE(8, 4)
    DR_CONV_B_CASE(16, 1)
    DR_CONV_B_CASE(16, 2)
    DR_CONV_B_CASE(16, 4)
#undef DR_CONV_B_CASE
    fail(DR_ERR_ARG, "launch_conv: no bf16x3 instance CI=%d CT=%d", c.ci, c.ct);
  }
#endif
#ifndef DR_PARITY_HOOKS  // the fused-skip forms (FeatureNet's stage-3 head in its literal order) are instantiated in the parity build only
  if (c.fz) fail(DR_ERR_UNSUPPORTED, "launch_conv: fused-skip kernels are built with -DDR_PARITY_HOOKS only");
#else
  if (c.async == 2 && c.fz) {
    if (c.fz != 8 || c.ci != 16 || c.nup != 12 || c.ct != 1 || c.ncw != 8) fail(DR_ERR_ARG, "launch_conv: no fused-skip marching instance FZ=%d CI=%d NUP=%d CT=%d", c.fz, c.ci, c.nup, c.ct);
    if (c.pt == 4) launch_conv_m_inst<16, 12, 1, 4, 8>(c, st);
    else launch_conv_m_inst<16, 12, 1, 2, 8>(c, st);
    return;
  }
#endif
  if (c.async == 2 && c.march.wino) {
#define DR_X(CI_, NUP_, CT_, PT_, NCW_) \
  if (c.ci == CI_ && c.nup == NUP_ && c.ct == CT_ && c.pt == PT_ && c.ncw == NCW_) { launch_conv_m_inst<CI_, NUP_, CT_, PT_, 0, NCW_, 1>(c, st); return; }
    DR_MARCH_W_INSTANCES(DR_X)
#undef DR_X
    fail(DR_ERR_ARG, "launch_conv: no Winograd marching instance CI=%d NUP=%d CT=%d PT=%d waves=%d", c.ci, c.nup, c.ct, c.pt, c.ncw);
  }
  if (c.async == 2) {
#define DR_X(CI_, NUP_, CT_, PT_, NCW_) \
  if (c.ci == CI_ && c.nup == NUP_ && c.ct == CT_ && c.pt == PT_ && c.ncw == NCW_) { launch_conv_m_inst<CI_, NUP_, CT_, PT_, 0, NCW_>(c, st); return; }
    DR_MARCH_INSTANCES(DR_X)
#undef DR_X
    fail(DR_ERR_ARG, "launch_conv: no marching instance CI=%d NUP=%d CT=%d PT=%d waves=%d", c.ci, c.nup, c.ct, c.pt, c.ncw);
  }
  if (c.async == 4) {
#define DR_CONV_W_CASE(CI_, CT_, PT_) if (c.ci == CI_ && c.ct == CT_ && c.pt == PT_) { launch_conv_w_inst<CI_, CT_, PT_>(c, st); return; }
    DR_CONV_W_CASE(16, 1, 1) DR_CONV_W_CASE(16, 1, 2) DR_CONV_W_CASE(16, 2, 1) DR_CONV_W_CASE(16, 2, 2)
    DR_CONV_W_CASE(8, 1, 1) DR_CONV_W_CASE(8, 1, 2)
#undef DR_CONV_W_CASE
    fail(DR_ERR_ARG, "launch_conv: no Winograd instance CI=%d CT=%d PT=%d", c.ci, c.ct, c.pt);
  }
  if (c.async) {
#define DR_CONV_A_CASE(CI_, CT_)                                                \
  if (c.ci == CI_ && c.ct == CT_) {                                             \
    if (c.pt == 4) launch_conv_a_inst<CI_, CT_, 4>(c, st);                      \
    else if (c.pt == 2) launch_conv_a_inst<CI_, CT_, 2>(c, st);                 \
    else launch_conv_a_inst<CI_, CT_, 1>(c, st);                                \
    return;                                                                     \
  }
    if (c.ci == 4 && c.ct == 1 && c.pt == 1) fail(DR_ERR_ARG, "launch_conv: no async instance CI=4 PT=1");
    DR_CONV_A_CASE(4, 1)
    DR_CONV_A_CASE(8, 1)
    DR_CONV_A_CASE(8, 2)
    DR_CONV_A_CASE(16, 1)
    DR_CONV_A_CASE(16, 2)
#undef DR_CONV_A_CASE
    fail(DR_ERR_ARG, "launch_conv: no async instance CI=%d CT=%d PT=%d", c.ci, c.ct, c.pt);
  }
#ifdef DR_PARITY_HOOKS
  if (c.fz) {
    if (c.fz != 8 || c.ci != 16 || c.ct != 1) fail(DR_ERR_ARG, "launch_conv: no fused-skip instance FZ=%d CI=%d CT=%d", c.fz, c.ci, c.ct);
    if (c.pt == 4) launch_conv_inst<16, 1, 4, 8>(c, st);
    else launch_conv_inst<16, 1, 1, 8>(c, st);
    return;
  }
#endif
  if (c.args.class_loop > 0) {
#define DR_CONV_C_CASE(CI_, CT_)                                                \
  if (c.ci == CI_ && c.ct == CT_) {                                             \
    if (c.pt == 4) launch_conv_c_inst<CI_, CT_, 4>(c, st);                      \
    else launch_conv_c_inst<CI_, CT_, 1>(c, st);                                \
    return;                                                                     \
  }
    DR_CONV_C_CASE(8, 1)
    DR_CONV_C_CASE(8, 2)
    DR_CONV_C_CASE(16, 1)
    DR_CONV_C_CASE(16, 2)
#undef DR_CONV_C_CASE
    fail(DR_ERR_ARG, "launch_conv: no class-loop instance CI=%d CT=%d PT=%d", c.ci, c.ct, c.pt);
  }
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
