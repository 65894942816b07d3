// conv_march.h -- k_conv_m: the stride-1 3x3 / 3x3x3 convolutions as a MARCHING, producer/consumer-specialised kernel.
//
// Same implicit GEMM as conv_mfma.h (operand mapping, packed weights, swizzled LDS-DMA image, epilogue), different engine:
//   * a persistent workgroup = 8 or 12 CONSUMER waves (two / three per SIMD: ds_read_b128 + v_mfma_f32_16x16x4_f32 and nothing
//     else in the K loop) + 2 PRODUCER waves that issue every LDS-DMA (global_load_lds_dwordx4) from a per-lane offset table built
//     once per workgroup -- no address decode, no vmcnt wait and no barrier in the MFMA waves;
//   * the workgroup owns a contiguous range of steps (march_plan.h) and MARCHES along z: a ring of R input planes stays
//     in LDS, a step computes one output plane of the tile from the three planes around it and only ONE new plane is
//     fetched per step (k_conv_a re-fetched a 6-plane halo per 4 output planes, in two half-record passes);
//   * no s_barrier after start-up.  Producer -> consumers: a monotone `ready` word per producer wave, written after the
//     wave's s_waitcnt vmcnt(0) (its DMA pieces have landed).  Consumers -> producer: one monotone `released` word per
//     consumer wave, written after the last ds_read of a ring slot.  A wave's LDS operations execute in order, so a
//     reader that has seen the word sees the data, and a slot is overwritten only after every reader is done with it.
//     The consumer waves therefore drift apart instead of draining the MFMA pipe at a barrier once per unit;
//   * sections whose input plane lies outside the volume (z padding) are skipped, not fed zeros;
//   * channel passes that do not fit (Cin = 32 with a 3x3x3 kernel: 72 KB of weights) run as an OUTER loop over the whole
//     range: pass 0 leaves raw fp32 partial sums in the output tensor, the last pass adds them before the BN/ReLU
//     epilogue (the same lane wrote them).
// Every wait is bounded: a wave that gives up raises the workgroup's abort word and *err and leaves; nothing can hang.
#pragma once
#include "march_plan.h"

namespace dr {

struct MarchArgs {
  const int *tap2d;    // [NUP * TPC] in-plane tap offsets (positions of the staged plane)
  MarchGeom geo;
  int NPO;             // channel passes around the whole range (1 or 2)
  int ncols, colsH, colsW;
  int R, PS;           // ring slots, 16-byte LDS slots per ring slot (multiple of 128: an even number of 1 KiB DMA pieces)
  int NP;              // staged positions per plane (TYI * TXI)
  int nit;             // DMA pieces of a plane per producer wave (PS / 128)
  int wsec;            // float4 per weight section (NUP * CT * 64)
  int NU;              // K chunks per channel pass in the packed weight array (KZ * NUP)
  int steps;           // ncols * Dc
  int ncw;             // consumer waves (8, 10 or 12)
  // Addressing (elements): position (image zc, marching index z, in-plane row y, x) of the input lies at
  //   zc * i_sv + z * i_sz + y * i_sy + x * inC, the output likewise with o_*.
  //   3-D layer  (KZ = 3, rm = 0): zc = 0, z = depth plane, y = row          i_sz = H*W*C, i_sy = W*C
  //   2-D tiles  (KZ = 1, rm = 0): zc = image, z = 0, y = row                i_sv = H*W*C, i_sy = W*C
  //   ROW MARCH  (KZ = 3, rm = 1): zc = image, z = ROW, no in-plane y        i_sv = H*W*C, i_sz = W*C, i_sy = 0
  // The row march is the same ring protocol applied to a 2-D layer: a "plane" is one row strip of one image, a step produces one
  // output row strip from the three row strips around it (sections = ky), so no row of the y halo is ever fetched twice.
  int i_sv, i_sz, i_sy, o_sv, o_sz, o_sy;
  int depth;           // loads a producer wave keeps in flight (1: publish every load before issuing the next)
  int inHp;            // rows a plane has inside the tensor (inH, or 1 for the row march)
  int rm;              // row march
  int wino;            // 1: the y axis in Winograd F(2,3) form (march_consumer_w): a position is a row PAIR, a.sy = 2, the in-plane tap table lists the x taps only
  int *err;            // raised when a wait gave up (nullptr: not reported)
};

constexpr int kMarchProducers = 2;       // DMA producer waves (the fused-skip producers are four: kMarchFzProducers)
constexpr int kMarchMaxConsumers = 12;   // consumer waves: 8 (two per SIMD), 10 (640-wide 2-D rows are 20 or 10 position tiles) or 12 (three per SIMD)
constexpr int kMarchMaxIt = 24;          // DMA pieces per producer wave per plane (planes up to 48 KB)
constexpr int kMarchSpinLimit = 1 << 18; // polls before a wait gives up (tens of milliseconds)
// flag words (ints) behind the weights: [0..7] ready (one per producer wave), [8..19] released (one per consumer wave), [28] abort
constexpr int kMarchFlagInts = 32, kMarchReleased = 8, kMarchAbort = 28;

// the flag words are read and written with LDS instructions (ds_read / ds_write), never through flat addressing
typedef __attribute__((address_space(3))) volatile int march_flag_t;

__device__ inline int march_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- index arithmetic shared with the host emulation (tests/cpp/march_emul.hip), which checks it against a direct convolution ----
// staged position of output position j of position tile (wave, pt)
DR_HD inline int march_bpos(const ConvArgs &a, int wave, int pt, int PT, int j) {
  const int tau = wave * PT + pt, xt = tau % a.TXT, yt = tau / a.TXT;
  return yt * a.sy * a.TXI + (xt * 16 + j) * a.sx;  // (a.sy = 1, or 2 in the Winograd form: position tile yt starts at row 2 * yt of the staged plane)
}
// column -> tile origin (output positions) and, for 2-D layers, the image the column belongs to
DR_HD inline void march_tile_origin(const ConvArgs &a, const MarchArgs &m, int col, int &zc, int &py0, int &px0) {
  zc = col / (m.colsH * m.colsW);
  const int tl = col - zc * (m.colsH * m.colsW);
  py0 = (tl / m.colsW) * a.TY; px0 = (tl % m.colsW) * a.TXT * 16;
}
// DMA piece (producer wave pw, iteration it), lane -> element offset inside the plane relative to the tile origin, packed (y, x)
template <int CI>
DR_HD inline void march_piece_entry(const ConvArgs &a, const MarchArgs &m, int pw, int it, int lane, int &rel, unsigned &yx) {
  int pos, c4;
  conv_a_slot<CI>((it * kMarchProducers + pw) * 64 + lane, pos, c4);
  const unsigned y = (unsigned)pos / (unsigned)a.TXI, x = (unsigned)pos - y * a.TXI;
  const bool ok = it < m.nit && pos < m.NP;
  rel = ok ? (int)(y * m.i_sy + x * a.inC + c4 * 4) : 0;
  yx = ok ? ((y << 16) | x) : 0x7fff7fffu;  // a position no tile origin can bring inside the tensor
}
DR_HD inline bool march_piece_inside(const ConvArgs &a, const MarchArgs &m, unsigned yx, int iy0, int ix0) {
  const unsigned gy = (unsigned)(iy0 + (int)(yx >> 16)), gx = (unsigned)(ix0 + (int)(yx & 0xffffu));
  return gy < (unsigned)m.inHp && gx < (unsigned)a.inW;
}
// element offset of the tile origin of input plane `plane` of image zc, channel slice `pass` (may be negative: the halo starts outside the tensor)
DR_HD inline long long march_plane_offset(const ConvArgs &a, const MarchArgs &m, int zc, int plane, int iy0, int ix0, int pass, int CI) {
  (void)CI;
  return (long long)zc * m.i_sv + (long long)plane * m.i_sz + (long long)iy0 * m.i_sy + (long long)ix0 * a.inC + (long long)pass * a.pass_stride;
}
// weight piece e = (sec * NUP + u) * CT + ct of outer pass po -> float4 index (lane 0) in the packed weight array
DR_HD inline size_t march_weight_src(const ConvArgs &a, const MarchArgs &m, int po, int e, int NUP, int CT, int ct0) {
  const int ct = e % CT, su = e / CT, u = su % NUP, sec = su / NUP;
  const int dz = sec / m.geo.NPI, pi = sec - dz * m.geo.NPI;
  return ((size_t)((po * m.geo.NPI + pi) * m.NU + dz * NUP + u) * a.ctTot + ct0 + ct) * 64;
}
DR_HD inline size_t march_out_index(const ConvArgs &a, const MarchArgs &m, int zc, int z, int qy, int qx, int c0) {
  return (size_t)zc * m.o_sv + (size_t)z * m.o_sz + (size_t)qy * m.o_sy + (size_t)qx * a.outC + c0;
}

// Consumer side: wait until load `idx` has landed.  `cached` remembers the last value seen (the producers normally run
// ahead, so most sections need no LDS read at all).
template <int NPW>
__device__ inline bool march_wait_ready(march_flag_t *flags, int idx, int &cached, int *err, int lane) {
  if (cached > idx) return true;
  for (int spin = 0; spin < kMarchSpinLimit; ++spin) {
    int r = flags[0];
#pragma unroll
    for (int p = 1; p < NPW; ++p) { const int t = flags[p]; r = t < r ? t : r; }
    cached = march_uniform(r);
    if (cached > idx) return true;
    if (march_uniform(flags[kMarchAbort])) break;
    __builtin_amdgcn_s_sleep(1);
  }
  if (lane == 0) { flags[kMarchAbort] = 1; if (err) *err = 1; }
  return false;
}
// Producer side: wait until every consumer wave has released `need` loads.
__device__ inline bool march_wait_released(march_flag_t *flags, int ncw, int need, int *err, int lane) {
  for (int spin = 0; spin < kMarchSpinLimit; ++spin) {
    int v = flags[kMarchReleased];
#pragma unroll
    for (int w = 1; w < kMarchMaxConsumers; ++w) { const int t = flags[kMarchReleased + (w < ncw ? w : 0)]; v = t < v ? t : v; }
    if (march_uniform(v) >= need) return true;
    if (march_uniform(flags[kMarchAbort])) break;
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) { flags[kMarchAbort] = 1; if (err) *err = 2; }
  return false;
}

// One poll of the same quantity (no waiting).
__device__ inline int march_released_now(march_flag_t *flags, int ncw) {
  int v = flags[kMarchReleased];
#pragma unroll
  for (int w = 1; w < kMarchMaxConsumers; ++w) { const int t = flags[kMarchReleased + (w < ncw ? w : 0)]; v = t < v ? t : v; }
  return march_uniform(v);
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate; the counter has six bits).
__device__ inline void march_wait_vmcnt(int n) {
  switch (n) {
#define DR_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    DR_VM(1) DR_VM(2) DR_VM(3) DR_VM(4) DR_VM(5) DR_VM(6) DR_VM(7) DR_VM(8) DR_VM(9) DR_VM(10) DR_VM(11) DR_VM(12) DR_VM(13) DR_VM(14) DR_VM(15) DR_VM(16)
    DR_VM(17) DR_VM(18) DR_VM(19) DR_VM(20) DR_VM(21) DR_VM(22) DR_VM(23) DR_VM(24) DR_VM(25) DR_VM(26) DR_VM(27) DR_VM(28) DR_VM(29) DR_VM(30) DR_VM(31)
    DR_VM(32) DR_VM(33) DR_VM(34) DR_VM(35) DR_VM(36) DR_VM(37) DR_VM(38) DR_VM(39) DR_VM(40) DR_VM(41) DR_VM(42) DR_VM(43) DR_VM(44) DR_VM(45) DR_VM(46)
    DR_VM(47) DR_VM(48) DR_VM(49) DR_VM(50) DR_VM(51) DR_VM(52) DR_VM(53) DR_VM(54) DR_VM(55) DR_VM(56) DR_VM(57) DR_VM(58) DR_VM(59) DR_VM(60)
#undef DR_VM
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// ---- K loop of one section: NUP chunks, fully unrolled, every LDS address known before the loop ----
template <int NUP, int CT, int PT>
__device__ inline void march_load(const float4 *tile, const float4 *wp, const int (&sw)[NUP][PT], int u, float4 (&av)[CT], float4 (&bv)[PT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) av[ct] = wp[(u * CT + ct) * 64];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) bv[pt] = tile[sw[u][pt]];
}
template <int CT, int PT>
__device__ inline void march_anchor(const float4 (&av)[CT], const float4 (&bv)[PT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) asm volatile("" ::"v"(av[ct].x), "v"(av[ct].y), "v"(av[ct].z), "v"(av[ct].w));
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) asm volatile("" ::"v"(bv[pt].x), "v"(bv[pt].y), "v"(bv[pt].z), "v"(bv[pt].w));
}
// DEPTH = how many chunks ahead the operands are fetched (DEPTH + 1 register sets used in rotation).  One chunk ahead
// leaves an LDS read 8 MFMAs = 256 cycles to return at CT = 1, PT = 2; with eight waves reading it takes longer than that
// (tools/ubench/march_kloop.hip), so the wave stalls at every chunk.
#ifndef DR_MARCH_DEPTH
#define DR_MARCH_DEPTH 2
#endif
template <int NUP, int CT, int PT, int DEPTH = (CT * PT >= 4 ? 1 : (CT * PT == 1 ? DR_MARCH_DEPTH + 1 : DR_MARCH_DEPTH))>  // 16 MFMAs per chunk cover an LDS round trip with one chunk of prefetch
__device__ inline void march_kloop(const float4 *tile, const float4 *wp, const int (&sw)[NUP][PT], floatx4 (&acc)[CT][PT]) {
  constexpr int NS = DEPTH + 1;
  float4 av[NS][CT], bv[NS][PT];
  // one accumulator per wave (CT = PT = 1): odd chunks go to a second one, so that consecutive MFMAs never wait for each
  // other (40-cycle dependent latency against a 32-cycle issue interval); the two are added at the end of the section
  floatx4 acc2[CT][PT];
  if constexpr (CT * PT == 1) acc2[0][0] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < DEPTH && u < NUP; ++u) march_load<NUP, CT, PT>(tile, wp, sw, u, av[u % NS], bv[u % NS]);
#pragma unroll
  for (int u = 0; u < NUP; ++u) {
    if (u + DEPTH < NUP) march_load<NUP, CT, PT>(tile, wp, sw, u + DEPTH, av[(u + DEPTH) % NS], bv[(u + DEPTH) % NS]);
    __builtin_amdgcn_sched_barrier(0);
    if (CT * PT == 1 && (u & 1)) conv_chunk_mfma<CT, PT>(av[u % NS], bv[u % NS], acc2);
    else conv_chunk_mfma<CT, PT>(av[u % NS], bv[u % NS], acc);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < NUP) march_anchor<CT, PT>(av[(u + 1) % NS], bv[(u + 1) % NS]);  // the next chunk's operands are waited for here, behind this chunk's MFMAs
  }
  if constexpr (CT * PT == 1) acc[0][0] += acc2[0][0];
}

// The same loop over ALL THREE sections of a step (KZ = 3, one channel slice per plane, no z padding in this step): the operand
// prefetch runs across the section boundaries, so the pipeline fills once per step instead of once per section (a section of
// 9-12 chunks otherwise starts with a bare LDS round trip: ~15 % of its MFMA time).  tile[s] / wp + s * wsec are the ring slot and
// the weight section of section s; `release(s)` is called after the last MFMA that reads section s.
template <int NUP, int CT, int PT, int DEPTH, class Release>
__device__ inline void march_kloop3(const float4 *const (&tile)[3], const float4 *wp, int wsec, const int (&sw)[NUP][PT], floatx4 (&acc)[CT][PT], Release release) {
  constexpr int NS = DEPTH + 1, NT = 3 * NUP;
  float4 av[NS][CT], bv[NS][PT];
  floatx4 acc2[CT][PT];
  if constexpr (CT * PT == 1) acc2[0][0] = floatx4{0.f, 0.f, 0.f, 0.f};
  const float4 *wps[3] = {wp, wp + wsec, wp + 2 * (size_t)wsec};
#pragma unroll
  for (int t = 0; t < DEPTH && t < NT; ++t) march_load<NUP, CT, PT>(tile[t / NUP], wps[t / NUP], sw, t % NUP, av[t % NS], bv[t % NS]);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t + DEPTH < NT) march_load<NUP, CT, PT>(tile[(t + DEPTH) / NUP], wps[(t + DEPTH) / NUP], sw, (t + DEPTH) % NUP, av[(t + DEPTH) % NS], bv[(t + DEPTH) % NS]);
    __builtin_amdgcn_sched_barrier(0);
    if (CT * PT == 1 && (t & 1)) conv_chunk_mfma<CT, PT>(av[t % NS], bv[t % NS], acc2);
    else conv_chunk_mfma<CT, PT>(av[t % NS], bv[t % NS], acc);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < NT) march_anchor<CT, PT>(av[(t + 1) % NS], bv[(t + 1) % NS]);
    // every read of section (t - DEPTH) / NUP has returned once the operands of chunk t + 1 (fetched DEPTH chunks ago) are in
    if (t % NUP == NUP - 1) release(t / NUP);
  }
  if constexpr (CT * PT == 1) acc[0][0] += acc2[0][0];
}

// ---- epilogue of one step; raw: 0 = final, 1 = store raw partial sums, 2 = add the stored partial sums, then final ----
template <int CT, int PT>
__device__ inline void march_epilogue(const ConvArgs &a, const MarchArgs &m, floatx4 (&acc)[CT][PT], const float4 (&scv)[CT], const float4 (&biv)[CT], int raw,
                                      int wave, int j, int g, int ct0, int zc, int z, int py0, int px0) {
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int tau = wave * PT + pt;
    const int xt = tau % a.TXT, yt = tau / a.TXT;
    const int qy = py0 + yt, qx = px0 + xt * 16 + j;
    if (qy >= a.nPH || qx >= a.nPW) continue;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c0 = (ct0 + ct) * 16 + 4 * g;
      if (c0 >= a.rows_valid) continue;
      const size_t obase = march_out_index(a, m, zc, z, qy, qx, c0);
      float4 v = make_float4(acc[ct][pt][0], acc[ct][pt][1], acc[ct][pt][2], acc[ct][pt][3]);
      if (raw == 2) {
        const float4 r = *reinterpret_cast<const float4 *>(a.out + obase);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (raw != 1) {
        const float4 sc = scv[ct], bi = biv[ct];
        v.x = v.x * sc.x + bi.x; v.y = v.y * sc.y + bi.y; v.z = v.z * sc.z + bi.z; v.w = v.w * sc.w + bi.w;
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.add_mode) {
          size_t abase = obase;
          if (a.add_mode == 2) abase = (((size_t)(zc + z) * a.addH + (qy >> 1)) * a.addW + (qx >> 1)) * a.outC + c0;  // (never the row march: one of zc, z is 0)
          const float4 r = *reinterpret_cast<const float4 *>(a.add + abase);
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
      }
      *reinterpret_cast<float4 *>(a.out + obase) = v;
    }
  }
}

template <int CI, int NUP, int CT, int PT, int NPW>
__device__ inline void march_consumer(const ConvArgs &a, const MarchArgs &m, float4 *lds4, march_flag_t *flags, const float4 *wl, int wave, int lane,
                                      int s0, int s1) {
  constexpr int TPC = 16 / CI;
  const int j = lane & 15, g = lane >> 4;
  const int sub = (4 * g) / CI, c4 = ((4 * g) % CI) / 4, ct0 = blockIdx.z * CT;
  // swizzled 16-byte slot of this lane's operand for every chunk of a plane and every position tile of the wave
  int sw[NUP][PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int bpos = march_bpos(a, wave, pt, PT, j);
#pragma unroll
    for (int u = 0; u < NUP; ++u) {
      sw[u][pt] = conv_a_unit<CI>(bpos + m.tap2d[u * TPC + sub], c4);
      asm volatile("" : "+v"(sw[u][pt]));  // one register per address: hipcc otherwise keeps the position and channel parts apart (2 x NUP x PT registers)
    }
  }
  float4 scv[CT], biv[CT];
  conv_load_affine<CT>(a, g, ct0, scv, biv);
  const float4 *wp = wl + lane;
  const int KZ = m.geo.KZ, NPI = m.geo.NPI, R = m.R, Dc = m.geo.Dc;
  int cached = 0, L = 0;
  for (int po = 0; po < m.NPO; ++po) {
    const int raw = m.NPO == 1 ? 0 : (po == 0 ? 1 : 2);  // the planner produces NPO <= 2
    for (int s = s0; s < s1;) {
      const MarchSeg sg = march_segment(m.geo, s, s1);
      int zc, py0, px0;
      march_tile_origin(a, m, sg.col, zc, py0, px0);
      MarchCursor cur;
      cur.begin(m.geo, sg, L, R);
      for (int z = sg.za; z < sg.zb; ++z) {
        floatx4 acc[CT][PT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
        int rel = cur.rel0, slot = cur.slot0;
#if !defined(DR_MARCH_NO_FLAT) && !defined(DR_MABL_NO_KLOOP)
        if (KZ == 3 && NPI == 1 && z > 0 && z < Dc - 1) {  // all three planes exist: one pipelined pass over the 3 * NUP chunks
          const int idx0 = L + rel;
#if !defined(DR_MABL_NO_WAIT) && !defined(DR_MABL_FREE)
          if (!march_wait_ready<NPW>(flags, idx0 + 2, cached, m.err, lane)) return;  // loads are published in order
#endif
          asm volatile("" ::: "memory");
          const int sl1 = slot + 1 >= R ? slot + 1 - R : slot + 1, sl2 = slot + 2 >= R ? slot + 2 - R : slot + 2;
          const float4 *const tiles[3] = {lds4 + (size_t)slot * m.PS, lds4 + (size_t)sl1 * m.PS, lds4 + (size_t)sl2 * m.PS};
          const bool last = z == sg.zb - 1;
          march_kloop3<NUP, CT, PT, (CT * PT >= 4 ? 1 : (CT * PT == 1 ? DR_MARCH_DEPTH + 1 : DR_MARCH_DEPTH))>(
              tiles, wp, m.wsec, sw, acc, [&](int sec) {
                if (sec == 0 || last) {
                  asm volatile("" ::: "memory");
                  if (lane == 0) flags[kMarchReleased + wave] = idx0 + sec + 1;
                }
              });
        } else
#endif
        for (int dz = 0; dz < KZ; ++dz) {
          const int plane = KZ == 3 ? z - 1 + dz : z;
          const bool there = plane >= 0 && plane < Dc;  // else: z padding, the section is skipped
          const bool rls = KZ == 1 || dz == 0 || z == sg.zb - 1;
          for (int pi = 0; pi < NPI; ++pi, ++rel, slot = slot + 1 == R ? 0 : slot + 1) {
            if (!there) continue;
            const int idx = L + rel;
#if !defined(DR_MABL_NO_WAIT) && !defined(DR_MABL_FREE)  // (timing-ablation builds of round 3: results are wrong by design)
            if (!march_wait_ready<NPW>(flags, idx, cached, m.err, lane)) return;
#endif
            asm volatile("" ::: "memory");
#ifndef DR_MABL_NO_KLOOP
            march_kloop<NUP, CT, PT>(lds4 + (size_t)slot * m.PS, wp + (size_t)(dz * NPI + pi) * m.wsec, sw, acc);
#endif
            if (rls) {
              asm volatile("" ::: "memory");
              if (lane == 0) flags[kMarchReleased + wave] = idx + 1;
            }
          }
        }
#if defined(DR_MABL_NO_EPI) || defined(DR_MABL_FREE)
        if (m.NPO > 7)  // never true: keeps the accumulators live
#endif
        march_epilogue<CT, PT>(a, m, acc, scv, biv, raw, wave, j, g, ct0, zc, z, py0, px0);
        cur.next_step(m.geo, R);
      }
      L += sg.nl;
      s += sg.zb - sg.za;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The Winograd F(2,3) form of the marching consumer (conv_wino.h has the algebra): a position tile is 16 x by one ROW PAIR, a step
// computes two output rows per position from the four rows around them with 4 products per (x tap, plane, channel) instead of 6.
// What makes it fit the marching kernel's LDS budget is that the TRANSFORMED weights are never stored: the ring's weight area holds the
// raw kernel rows g0, g1 / 2, g2 (the direct form's 36 KB for a 16-channel XPAIR layer -- four transformed fragments would be 48 KB and,
// with a three-slot ring of 40 KB planes, 172 KB), and every lane derives u1 = (g0 + g2) / 2 + g1 / 2 and u2 = (g0 + g2) / 2 - g1 / 2
// for its fragment in registers (12 VALU per chunk, beside the 16 of the input transform, under 16 MFMAs = 512 cycles).
// Weight chunk order inside a section: [chunk of x taps r][kernel row k = 0..2][row tile]; NUP = 3 * NRP as in the direct form.
template <int NRP, int CT, int PT>
struct MarchWSet {  // operands of one chunk: three raw weight fragments per row tile, four raw input rows per position tile
  float4 g[3][CT], d[4][PT];
};
template <int NRP, int CT, int PT>
__device__ inline void march_w_load(const float4 *tile, const float4 *wsec, const int (&sw)[NRP][4][PT], int r, MarchWSet<NRP, CT, PT> &o) {
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) o.g[k][ct] = wsec[((r * 3 + k) * CT + ct) * 64];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int q = 0; q < 4; ++q) o.d[q][pt] = tile[sw[r][q][pt]];
}
template <int NRP, int CT, int PT>
__device__ inline void march_w_anchor(const MarchWSet<NRP, CT, PT> &o) {
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) asm volatile("" ::"v"(o.g[k][ct].x), "v"(o.g[k][ct].y), "v"(o.g[k][ct].z), "v"(o.g[k][ct].w));
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) asm volatile("" ::"v"(o.d[q][pt].x), "v"(o.d[q][pt].y), "v"(o.d[q][pt].z), "v"(o.d[q][pt].w));
}
DR_HD inline float march_w_u1(float g0, float g1h, float g2) { return fmaf(g0 + g2, 0.5f, g1h); }   // shared with the host emulation
DR_HD inline float march_w_u2(float g0, float g1h, float g2) { return fmaf(g0 + g2, 0.5f, -g1h); }
template <int NRP, int CT, int PT>
__device__ inline void march_w_compute(MarchWSet<NRP, CT, PT> &o, floatx4 (&acc)[4][CT][PT]) {
  float4 u[4][CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float4 g0 = o.g[0][ct], gh = o.g[1][ct], g2 = o.g[2][ct];
    u[0][ct] = g0; u[3][ct] = g2;
#ifdef DR_WABL_NO_WT  // timing ablation (results wrong by design): what the per-lane weight transform costs
    u[1][ct] = gh; u[2][ct] = gh;
#else
    u[1][ct] = make_float4(march_w_u1(g0.x, gh.x, g2.x), march_w_u1(g0.y, gh.y, g2.y), march_w_u1(g0.z, gh.z, g2.z), march_w_u1(g0.w, gh.w, g2.w));
    u[2][ct] = make_float4(march_w_u2(g0.x, gh.x, g2.x), march_w_u2(g0.y, gh.y, g2.y), march_w_u2(g0.z, gh.z, g2.z), march_w_u2(g0.w, gh.w, g2.w));
#endif
  }
#ifndef DR_WABL_NO_IT  // timing ablation: what the input transform costs
  conv_w_transform<PT>(o.d);
#endif
  conv_w_mfma<CT, PT>(u, o.d, acc);
}

template <int CT, int PT>
__device__ inline void march_epilogue_w(const ConvArgs &a, const MarchArgs &m, floatx4 (&acc)[4][CT][PT], const float4 (&scv)[CT], const float4 (&biv)[CT], int raw,
                                        int wave, int j, int g, int ct0, int zc, int z, int py0, int px0) {
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int tau = wave * PT + pt;
    const int xt = tau % a.TXT, yt = tau / a.TXT;
    const int qy = py0 + yt, qx = px0 + xt * 16 + j;
    if (qy >= a.nPH || qx >= a.nPW) continue;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c0 = (ct0 + ct) * 16 + 4 * g;
      if (c0 >= a.rows_valid) continue;
      const floatx4 m0 = acc[0][ct][pt], m1 = acc[1][ct][pt], m2 = acc[2][ct][pt], m3 = acc[3][ct][pt];
      const floatx4 o[2] = {(m0 + m1) + m2, (m1 - m2) - m3};
      // both rows' partial sums / residual operands are fetched before the first store (they may alias `out`: hipcc would keep each load behind the store before it)
      size_t obase[2];
      float4 ps[2], ad[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int oy = 2 * qy + r;
        obase[r] = march_out_index(a, m, zc, z, oy, qx, c0);
        ps[r] = ad[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (raw == 2) ps[r] = *reinterpret_cast<const float4 *>(a.out + obase[r]);
        if (raw != 1 && a.add_mode) {
          size_t abase = obase[r];
          if (a.add_mode == 2) abase = (((size_t)(zc + z) * a.addH + (oy >> 1)) * a.addW + (qx >> 1)) * a.outC + c0;
          ad[r] = *reinterpret_cast<const float4 *>(a.add + abase);
        }
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float4 v = make_float4(o[r][0], o[r][1], o[r][2], o[r][3]);
        if (raw == 2) { v.x += ps[r].x; v.y += ps[r].y; v.z += ps[r].z; v.w += ps[r].w; }
        if (raw != 1) {
          const float4 sc = scv[ct], bi = biv[ct];
          v.x = v.x * sc.x + bi.x; v.y = v.y * sc.y + bi.y; v.z = v.z * sc.z + bi.z; v.w = v.w * sc.w + bi.w;
          if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (a.add_mode) { v.x += ad[r].x; v.y += ad[r].y; v.z += ad[r].z; v.w += ad[r].w; }
        }
        *reinterpret_cast<float4 *>(a.out + obase[r]) = v;
      }
    }
  }
}

// 3-D layers only (KZ = 3, one channel slice per plane: NPI = 1).  The sections of a step that exist (planes inside the volume) are a
// contiguous range [lo, hi); their chunks run as ONE software pipeline (operands of the next chunk, of this or the next section, fetched
// under the MFMAs of the current one), every index a compile-time constant so that the two operand sets stay in registers.
template <int CI, int NUP, int CT, int PT, int NPW>
__device__ inline void march_consumer_w(const ConvArgs &a, const MarchArgs &m, float4 *lds4, march_flag_t *flags, const float4 *wl, int wave, int lane,
                                        int s0, int s1) {
  static_assert(NUP % 3 == 0, "three kernel rows per chunk of x taps");
  constexpr int TPC = 16 / CI, NRP = NUP / 3;
  const int j = lane & 15, g = lane >> 4;
  const int sub = (4 * g) / CI, c4 = ((4 * g) % CI) / 4, ct0 = blockIdx.z * CT;
  int sw[NRP][4][PT];  // swizzled 16-byte slot of this lane's operand: chunk of x taps, row of the pair's four-row window, position tile
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int bpos = march_bpos(a, wave, pt, PT, j);
#pragma unroll
    for (int r = 0; r < NRP; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sw[r][q][pt] = conv_a_unit<CI>(bpos + m.tap2d[r * TPC + sub] + q * a.TXI, c4);
        asm volatile("" : "+v"(sw[r][q][pt]));
      }
  }
  float4 scv[CT], biv[CT];
  conv_load_affine<CT>(a, g, ct0, scv, biv);
  const float4 *wp = wl + lane;
  const int R = m.R, Dc = m.geo.Dc;
  int cached = 0, L = 0;
  for (int po = 0; po < m.NPO; ++po) {
    const int raw = m.NPO == 1 ? 0 : (po == 0 ? 1 : 2);
    for (int s = s0; s < s1;) {
      const MarchSeg sg = march_segment(m.geo, s, s1);
      int zc, py0, px0;
      march_tile_origin(a, m, sg.col, zc, py0, px0);
      MarchCursor cur;
      cur.begin(m.geo, sg, L, R);
      for (int z = sg.za; z < sg.zb; ++z) {
        floatx4 acc[4][CT][PT];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[p][ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
        const int lo = z > 0 ? 0 : 1, hi = z < Dc - 1 ? 3 : 2, nsec = hi - lo;  // planes z - 1 + dz inside the volume
        const int idx0 = L + cur.rel0;                                          // load index of section dz = 0 (whether it exists or not)
        if (!march_wait_ready<NPW>(flags, idx0 + hi - 1, cached, m.err, lane)) return;  // loads are published in order
        asm volatile("" ::: "memory");
        const bool last = z == sg.zb - 1;
        auto tile_of = [&](int i) {  // ring slot of the i-th existing section
          int sl = cur.slot0 + lo + i;
          sl = sl >= R ? sl - R : sl;
          return (const float4 *)(lds4 + (size_t)sl * m.PS);
        };
        auto wsec_of = [&](int i) { return wp + (size_t)(lo + i) * m.wsec; };
        MarchWSet<NRP, CT, PT> set[2];
        march_w_load<NRP, CT, PT>(tile_of(0), wsec_of(0), sw, 0, set[0]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (i < nsec) {
#pragma unroll
            for (int r = 0; r < NRP; ++r) {
              const int t = i * NRP + r;  // compile-time: the sets alternate
              if (r + 1 < NRP) march_w_load<NRP, CT, PT>(tile_of(i), wsec_of(i), sw, r + 1, set[(t + 1) & 1]);
              else if (i + 1 < nsec) march_w_load<NRP, CT, PT>(tile_of(i + 1), wsec_of(i + 1), sw, 0, set[(t + 1) & 1]);
              __builtin_amdgcn_sched_barrier(0);
              march_w_compute<NRP, CT, PT>(set[t & 1], acc);
              __builtin_amdgcn_sched_barrier(0);
              if (r + 1 < NRP || i + 1 < nsec) march_w_anchor<NRP, CT, PT>(set[(t + 1) & 1]);
              // every read of section i has returned here (its last chunk's operands were anchored one chunk ago): release its slot if no later step reads it
              if (r == NRP - 1 && (lo + i == 0 || last)) {
                asm volatile("" ::: "memory");
                if (lane == 0) flags[kMarchReleased + wave] = idx0 + lo + i + 1;
              }
            }
          }
        }
        march_epilogue_w<CT, PT>(a, m, acc, scv, biv, raw, wave, j, g, ct0, zc, z, py0, px0);
        cur.next_step(m.geo, R);
      }
      L += sg.nl;
      s += sg.zb - sg.za;
    }
  }
}

// Producer wave pw of kMarchProducers: takes DMA pieces pw, pw + 2, ... of every plane and of the weights.
// Up to m.depth loads of the wave are in flight: after issuing the pieces of load i it waits only until the pieces of load
// i - depth + 1 have landed (s_waitcnt vmcnt(in flight behind it); a wave's loads return in order) and publishes THAT one.  Small planes
// (the row march: 10-20 KB per load, a step of 1-2 us) would otherwise be fetched one memory latency after the other.
// When the ring has no free slot the wave first drains and publishes everything it has in flight -- the consumers may need
// exactly those loads to release the slot it is waiting for.
template <int CI>
__device__ inline void march_producer(const ConvArgs &a, const MarchArgs &m, float4 *lds4, march_flag_t *flags, float4 *wl, int pw, int lane, int s0,
                                      int s1, int NUP, int CT) {
  const int ct0 = blockIdx.z * CT, NS = m.geo.KZ * m.geo.NPI;
  // per-lane table of this wave's pieces of a plane: element offset inside the plane (relative to the tile origin) and
  // the (y, x) of the position inside the tile for the bounds test.  Built once; a piece then costs a compare and an add.
  int rel[kMarchMaxIt];
  unsigned yx[kMarchMaxIt];
#pragma unroll
  for (int it = 0; it < kMarchMaxIt; ++it) march_piece_entry<CI>(a, m, pw, it, lane, rel[it], yx[it]);
  const int depth = m.depth;
  int L = 0;
  for (int po = 0; po < m.NPO; ++po) {
    if (po > 0 && !march_wait_released(flags, m.ncw, L, m.err, lane)) return;  // nobody reads the previous pass's weights any more
    // packed weights of this outer pass: section (dz, pi) = chunks dz*NUP .. of channel pass po*NPI + pi
    for (int e = pw; e < NS * NUP * CT; e += kMarchProducers)  // piece e = (sec * NUP + u) * CT + ct
      conv_a_dma16(a.wpk + march_weight_src(a, m, po, e, NUP, CT, ct0) + lane, march_uniform(conv_a_lds_addr(wl + (size_t)e * 64)));
    int flight = 0;  // loads issued and not yet published (they are idx - flight .. idx - 1)
    for (int s = s0; s < s1;) {
      const MarchSeg sg = march_segment(m.geo, s, s1);
      int zc, py0, px0;
      march_tile_origin(a, m, sg.col, zc, py0, px0);
      const int iy0 = m.rm ? 0 : py0 * a.sy - a.py, ix0 = px0 * a.sx - a.px;  // (row march: the plane IS the row, no y padding inside it)
      for (int l = 0; l < sg.nl; ++l) {
        const int idx = L + l;
        int plane, pi;
        march_load_plane(m.geo, sg, l, plane, pi);
#ifndef DR_MABL_FREE
        if (idx >= m.R && march_released_now(flags, m.ncw) < idx - m.R + 1) {
          if (flight) {
            conv_a_wait_dma();
            if (lane == 0) flags[pw] = idx;
            flight = 0;
          }
          if (!march_wait_released(flags, m.ncw, idx - m.R + 1, m.err, lane)) return;
        }
#endif
        asm volatile("" ::: "memory");
        const float *pbase = a.in + march_plane_offset(a, m, zc, plane, iy0, ix0, po * m.geo.NPI + pi, CI);
        float4 *dst = lds4 + (size_t)(idx % m.R) * m.PS;
#pragma unroll
        for (int it = 0; it < kMarchMaxIt; ++it) {
#if defined(DR_MABL_NO_DMA) || defined(DR_MABL_FREE)
          if (it < m.nit && m.NPO > 7) {
#else
          if (it < m.nit) {
#endif
            const float *src = march_piece_inside(a, m, yx[it], iy0, ix0) ? pbase + rel[it] : a.zero16;
            conv_a_dma16(src, march_uniform(conv_a_lds_addr(dst + (it * kMarchProducers + pw) * 64)));
          }
        }
        ++flight;
        if (flight >= depth) {  // the oldest load in flight (and, first time round, the weights) has landed once at most (flight - 1) * nit pieces are outstanding
          march_wait_vmcnt((flight - 1) * m.nit);
          if (lane == 0) flags[pw] = idx - flight + 2;
          --flight;
        }
      }
      L += sg.nl;
      s += sg.zb - sg.za;
    }
    if (flight) {
      conv_a_wait_dma();
      if (lane == 0) flags[pw] = L;
    }
  }
}

// Fused-skip producer (FeatureNet out.stage3, module.py:517-529): the layer's 32-channel input `inter3 = W1 . c3 + b1 +
// nearest_up2(inter2)` is never materialised -- producer wave pw computes channel group pw & 3 (4 channels) of the current
// 16-channel pass for every other batch of staged positions of the tile and writes it into the ring slot with ds_write_b128 (the image
// the DMA would have produced: same slot permutation, zeros outside the tensor).  Its 4 x 8 weights are wave-uniform
// (scalar registers); a lane reads 32 contiguous bytes of c3 and 16 bytes of inter2.  The arithmetic is k_skip_up's
// (same fmaf chain, (acc + b) + up), so the result is bit-identical to the two-kernel path.
constexpr int kMarchFzProducers = 8;  // 4 channel groups x 2 interleaved halves of the positions: one batch of loads in flight per wave and load
template <int FZ>
__device__ inline void march_producer_fz(const ConvArgs &a, const MarchArgs &m, float4 *lds4, march_flag_t *flags, float4 *wl, int pw, int lane, int s0,
                                         int s1, int NUP, int CT) {
  static_assert(FZ == 8, "8-channel skip source");
  constexpr int kB = 5;  // positions per lane whose loads are in flight together
  const int ct0 = blockIdx.z * CT, NS = m.geo.KZ * m.geo.NPI;
  const int Hc = a.inH >> 1, Wc = a.inW >> 1;
  for (int e = pw; e < NS * NUP * CT; e += kMarchFzProducers)
    conv_a_dma16(a.wpk + march_weight_src(a, m, 0, e, NUP, CT, ct0) + lane, march_uniform(conv_a_lds_addr(wl + (size_t)e * 64)));
  conv_a_wait_dma();
  int L = 0;
  for (int s = s0; s < s1;) {
    const MarchSeg sg = march_segment(m.geo, s, s1);
    int zc, py0, px0;
    march_tile_origin(a, m, sg.col, zc, py0, px0);
    const int iy0 = m.rm ? 0 : py0 * a.sy - a.py, ix0 = px0 * a.sx - a.px;
    for (int l = 0; l < sg.nl; ++l) {
      const int idx = L + l;
      int plane, pi;
      march_load_plane(m.geo, sg, l, plane, pi);
      if (idx >= m.R && !march_wait_released(flags, m.ncw, idx - m.R + 1, m.err, lane)) return;
      asm volatile("" ::: "memory");
      const int cg = pw & 3, half = pw >> 2;
      const int q = march_uniform(pi * 4 + cg);  // 4-channel group of inter3 this wave produces for this load
      float wr[4][FZ];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < FZ; ++c) wr[r][c] = a.fz_w[(4 * q + r) * FZ + c];
      const float4 fb = *reinterpret_cast<const float4 *>(a.fz_b + 4 * q);
      float4 *dst = lds4 + (size_t)(idx % m.R) * m.PS;
      for (int p0 = half * 64 * kB; p0 < m.NP; p0 += 2 * 64 * kB) {
        float4 xv[kB][FZ / 4], up[kB];
        bool in[kB];
#pragma unroll
        for (int k = 0; k < kB; ++k) {
          const int pos = p0 + k * 64 + lane;
          const unsigned y = (unsigned)pos / (unsigned)a.TXI, x = (unsigned)pos - y * a.TXI;
          const int gy = m.rm ? plane : iy0 + (int)y, gx = ix0 + (int)x;  // image row: the marching index in the row march
          in[k] = pos < m.NP && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW;
          up[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int c = 0; c < FZ / 4; ++c) xv[k][c] = up[k];
          if (in[k]) {
            const float *xp = a.fz_x + (((size_t)zc * a.inH + gy) * a.inW + gx) * FZ;
#pragma unroll
            for (int c = 0; c < FZ / 4; ++c) xv[k][c] = *reinterpret_cast<const float4 *>(xp + 4 * c);
            up[k] = *reinterpret_cast<const float4 *>(a.fz_coarse + (((size_t)zc * Hc + (gy >> 1)) * Wc + (gx >> 1)) * a.inC + 4 * q);
          }
        }
#pragma unroll
        for (int k = 0; k < kB; ++k) {
          const int pos = p0 + k * 64 + lane;
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
          if (pos < m.NP) dst[conv_a_unit<16>(pos, cg)] = o;
        }
      }
      asm volatile("" ::: "memory");  // the ds_writes above and the flag below execute in program order
      if (lane == 0) flags[pw] = idx + 1;
    }
    L += sg.nl;
    s += sg.zb - sg.za;
  }
}

// grid = (persistent workgroups (multiple of 8), 1, output-row groups); 8 consumer waves + 2 (DMA) or 4 (fused skip) producer waves.
template <int CI, int NUP, int CT, int PT, int FZ = 0, int NCW = 8, int W = 0>
__global__ __launch_bounds__(64 * (NCW + (FZ ? kMarchFzProducers : kMarchProducers))) void k_conv_m(const ConvArgs a, const MarchArgs m) {
  extern __shared__ float4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = march_uniform(tid >> 6);
  float4 *wl = lds4 + (size_t)m.R * m.PS;
  march_flag_t *flags = (march_flag_t *)(__attribute__((address_space(3))) char *)(wl + (size_t)m.geo.KZ * m.geo.NPI * m.wsec);
  if (tid < kMarchFlagInts) flags[tid] = 0;
  __syncthreads();  // the only barrier of the kernel
  // XCD k (= blockIdx.x % 8, own L2) owns the k-th contiguous eighth of the step space: the workgroups of an XCD work on
  // neighbouring columns, whose halos overlap in that L2
  const int nwg = gridDim.x, id = (int)(blockIdx.x & 7u) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  int s0, s1;
  march_range(m.steps, id, nwg, s0, s1);
  if (s0 >= s1) return;
  if (wave < NCW) {
    if constexpr (W) march_consumer_w<CI, NUP, CT, PT, kMarchProducers>(a, m, lds4, flags, wl, wave, lane, s0, s1);
    else march_consumer<CI, NUP, CT, PT, (FZ ? kMarchFzProducers : kMarchProducers)>(a, m, lds4, flags, wl, wave, lane, s0, s1);
  }
  else if constexpr (FZ > 0) march_producer_fz<FZ>(a, m, lds4, flags, wl, wave - NCW, lane, s0, s1, NUP, CT);
  else march_producer<CI>(a, m, lds4, flags, wl, wave - NCW, lane, s0, s1, NUP, CT);
}

}  // namespace dr
