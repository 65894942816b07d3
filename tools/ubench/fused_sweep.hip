// Go / no-go microbenchmark for fusing the plane sweep into k_conv_m's producers ("next round" item 1).
//
// One workgroup per CU, as k_conv_m<16,12,1,1,0,8,1> (s2.conv0, Winograd march): 8 CONSUMER waves run the real chunk loop of
// march_consumer_w (ds_read_b128 + input / weight transforms + v_mfma_f32_16x16x4_f32) on a three-slot ring of 40 KB planes, and NPW
// PRODUCER waves run the cost-volume iteration of k_costvol3<16> (cv_project / four 16-byte gathers per (pixel, view) from a bordered
// channels-last feature map / cv_warp / gate / accumulate, six source views) for the 630 staged positions of the NEXT plane of the tile
// and write it into a ring slot with ds_write_b128 -- the instruction mix a fused launch would have.  There is no hand-over protocol and
// nothing is checked: the question is only how the two kinds of waves slow each other down on one CU.
//   mode 1: consumers only      mode 2: producers only      mode 3: both, free-running for the same number of steps
// A "step" = one output plane of the tile = 3 sections x 4 chunks x 16 MFMAs per consumer wave, one 630-position plane per workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast fused_sweep.hip -o fused_sweep
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../tandem_amd/csrc/conv_mfma.h"
#include "../../tandem_amd/csrc/mvs_kernels.h"
namespace dr { std::string &last_error_slot() { static std::string s; return s; } }
using namespace dr;

struct SweepArgs {
  const float *feat;  // (7, h + 2, w + 2, 16), zero border
  float *out;
  float M[6][12];
  float gw[16];
  float gA1, gB1, gA2, gB2;
  int h, w, TXI, TYI, NPOS, D, PS, nsteps, mode, prio;
  float lo0, rng;
};

constexpr int NRP = 4, CT = 1, PT = 1, NCW = 8, C = 16;

__device__ inline void consumers(const SweepArgs &a, float4 *lds4, const float4 *wl, int wave, int lane) {
  const int j = lane & 15, g = lane >> 4;
  int sw[NRP][4][PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int bpos = (wave * PT + pt) * 2 * a.TXI + j * 2;  // row pair `wave` of the tile, XPAIR: positions two pixels apart
#pragma unroll
    for (int r = 0; r < NRP; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        sw[r][q][pt] = conv_a_unit<16>(bpos + r + q * a.TXI, g);
        asm volatile("" : "+v"(sw[r][q][pt]));
      }
  }
  const float4 *wp = wl + lane;
  const int wsec = NRP * 3 * CT * 64;
  floatx4 tot = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int step = 0; step < a.nsteps; ++step) {
    floatx4 acc[4][CT][PT];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p][0][0] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto tile_of = [&](int i) { return (const float4 *)(lds4 + (size_t)((step + i) % 3) * a.PS); };
    auto wsec_of = [&](int i) { return wp + (size_t)i * wsec; };
    MarchWSet<NRP, CT, PT> set[2];
    march_w_load<NRP, CT, PT>(tile_of(0), wsec_of(0), sw, 0, set[0]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int r = 0; r < NRP; ++r) {
        const int t = i * NRP + r;
        if (r + 1 < NRP) march_w_load<NRP, CT, PT>(tile_of(i), wsec_of(i), sw, r + 1, set[(t + 1) & 1]);
        else if (i + 1 < 3) march_w_load<NRP, CT, PT>(tile_of(i + 1), wsec_of(i + 1), sw, 0, set[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        march_w_compute<NRP, CT, PT>(set[t & 1], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (r + 1 < NRP || i + 1 < 3) march_w_anchor<NRP, CT, PT>(set[(t + 1) & 1]);
      }
    }
    // the output transform of the real epilogue, and a store of the tile's two rows (8 channels x 2 x: 16 bytes per lane and row)
    const floatx4 o0 = (acc[0][0][0] + acc[1][0][0]) + acc[2][0][0], o1 = (acc[1][0][0] - acc[2][0][0]) - acc[3][0][0];
    tot += o0 + o1;
    float4 *op = reinterpret_cast<float4 *>(a.out) + ((size_t)blockIdx.x * 64 + (step & 63)) * 1024 + (wave * 2) * 64 + lane;
    op[0] = make_float4(o0[0], o0[1], o0[2], o0[3]);
    op[64] = make_float4(o1[0], o1[1], o1[2], o1[3]);
  }
  if (tot[0] == 123.456f) a.out[0] = tot[1];
}

struct BatchCtx {
  CvProj P0, P1;
  float4 ref;
  int pos;
  bool inside;
};

template <int NPW, int NB>
__device__ inline void producers(const SweepArgs &a, float4 *lds4, const float *sM, int pw, int lane) {
  static_assert(6 % NB == 0, "the tap buffers rotate in step with the six views of a pixel batch");
  constexpr int LA = NB - 1;  // views whose gathers are in flight while one is consumed
  const int q = lane & 3, pix = lane >> 2;
  const int h = a.h, w = a.w, wp = w + 2;
  const size_t vplane = (size_t)(h + 2) * wp * C;
  const float fw = (float)w, fh = (float)h;
  const float4 gw = make_float4(a.gw[q * 4], a.gw[q * 4 + 1], a.gw[q * 4 + 2], a.gw[q * 4 + 3]);
  const float *f00 = a.feat + ((size_t)wp + 1) * C + q * 4;
  const int nb = (a.NPOS + 15) / 16;
  const int tiles_x = w / 32, tiles = tiles_x * (h / 16);
  struct Taps { float4 t00, t01, t10, t11; };
  Taps buf[NB];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int step = 0, b = pw;
  auto setup = [&](int st, int bb, BatchCtx &B) {
    const int col = ((int)blockIdx.x * 7 + st / a.D) % tiles, d = st % a.D;
    const int ty0 = (col / tiles_x) * 16 - 1, tx0 = (col % tiles_x) * 32 - 1;
    B.pos = bb * 16 + pix;
    const int py = B.pos / a.TXI, px = B.pos - py * a.TXI;
    const int gy = ty0 + py, gx = tx0 + px;
    B.inside = B.pos < a.NPOS && gy >= 0 && gy < h && gx >= 0 && gx < w;
    const int cy = min(max(gy, 0), h - 1), cx = min(max(gx, 0), w - 1);
    B.ref = ld4(a.feat + ((size_t)(cy + 1) * wp + cx + 1) * C + q * 4);
    const float lo = a.lo0 + 0.002f * (float)((cx * 7 + cy * 13) & 63);  // (the fused kernel reads the tile's hypothesis base from LDS)
    const float depth = __builtin_fmaf(a.rng, (float)d * (1.f / (float)a.D), lo);
    B.P0 = cv_project(sM + 12 * q, depth, (float)cx, (float)cy, fw, fh, wp, C);
    B.P1 = cv_project(sM + 12 * (4 + (q & 1)), depth, (float)cx, (float)cy, fw, fh, wp, C);
  };
  auto issue = [&](const BatchCtx &B, int v, Taps &T) {
    const CvProj &P = v < 4 ? B.P0 : B.P1;
    const int o = cv_bcast_i(P.o, 4, v < 4 ? v : v - 4);
    const float *r0 = f00 + (size_t)(v + 1) * vplane, *r1 = r0 + (size_t)wp * C;
    T.t00 = ld4(r0 + o); T.t01 = ld4(r0 + o + C); T.t10 = ld4(r1 + o); T.t11 = ld4(r1 + o + C);
  };
  auto consume = [&](const BatchCtx &B, int v, const Taps &T) {
    const CvProj &P = v < 4 ? B.P0 : B.P1;
    const int jj = v < 4 ? v : v - 4;
    CvTaps X;
    X.t00 = T.t00; X.t01 = T.t01; X.t10 = T.t10; X.t11 = T.t11;
    X.w00 = cv_bcast_f(P.w00, 4, jj); X.w01 = cv_bcast_f(P.w01, 4, jj); X.w10 = cv_bcast_f(P.w10, 4, jj); X.w11 = cv_bcast_f(P.w11, 4, jj);
    if (v == 0) acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 wv = cv_warp(X);
    const float4 df = make_float4(wv.x - B.ref.x, wv.y - B.ref.y, wv.z - B.ref.z, wv.w - B.ref.w);
    const float4 d2 = make_float4(df.x * df.x, df.y * df.y, df.z * df.z, df.w * df.w);
    float s = cv_gate_dot(gw, d2);
    s = cv_dpp_add(s, 0);
    s = cv_dpp_add(s, 1);
    const float g1 = fmaxf(__builtin_fmaf(a.gA1, s, a.gB1), 0.f);
    const float g = fmaxf(__builtin_fmaf(a.gA2, g1, a.gB2), 0.f) + 1.f;
    acc.x = __builtin_fmaf(g, d2.x, acc.x); acc.y = __builtin_fmaf(g, d2.y, acc.y); acc.z = __builtin_fmaf(g, d2.z, acc.z); acc.w = __builtin_fmaf(g, d2.w, acc.w);
  };
  BatchCtx cur, nxt;
  if (b >= nb) return;
  setup(step, b, cur);
#pragma unroll
  for (int k = 0; k < LA; ++k) issue(cur, k, buf[k % NB]);
  while (step < a.nsteps) {
    int nstep = step, nb2 = b + NPW;
    if (nb2 >= nb) { nb2 = pw; ++nstep; }
#pragma unroll
    for (int v = 0; v < 6; ++v) {
      const int vi = v + LA;
      if (vi == 6) setup(nstep, nb2, nxt);  // (past the end: one harmless extra batch)
      if (vi < 6) issue(cur, vi, buf[vi % NB]);
      else issue(nxt, vi - 6, buf[vi % NB]);
      consume(cur, v, buf[v % NB]);
    }
    const float rcp_n = 1.f / 6.f;
    float4 o4 = make_float4(acc.x * rcp_n, acc.y * rcp_n, acc.z * rcp_n, acc.w * rcp_n);
    if (!cur.inside) o4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cur.pos < a.NPOS) (lds4 + (size_t)((step + 2) % 3) * a.PS)[conv_a_unit<16>(cur.pos, q)] = o4;
    cur = nxt;
    step = nstep; b = nb2;
  }
}

template <int NPW, int NB>
__global__ __launch_bounds__(64 * (NCW + NPW)) void k_fused(const SweepArgs a) {
  extern __shared__ float4 lds4[];
  __shared__ float sM[6 * 12];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float4 *wl = lds4 + (size_t)3 * a.PS;
  for (int i = tid; i < 3 * a.PS + 3 * NRP * 3 * CT * 64; i += 64 * (NCW + NPW)) lds4[i] = make_float4((i & 255) * 1e-3f, 0.5f, 0.25f, 0.125f);
  for (int i = tid; i < 72; i += 64 * (NCW + NPW)) sM[i] = a.M[i / 12][i % 12];
  __syncthreads();
  if (wave < NCW) { if (a.prio) __builtin_amdgcn_s_setprio(3); if (a.mode & 1) consumers(a, lds4, wl, wave, lane); }  // prio: the MFMA waves win every issue arbitration
  else if (a.mode & 2) producers<NPW, NB>(a, lds4, sM, wave - NCW, lane);
}

template <int NPW, int NB>
static void run(SweepArgs a, const char *tag) {
  const size_t lds = ((size_t)3 * a.PS + 3 * NRP * 3 * CT * 64) * 16;
  auto kern = k_fused<NPW, NB>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float t[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    a.mode = mode;
    kern<<<256, 64 * (NCW + NPW), lds>>>(a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, 64 * (NCW + NPW), lds>>>(a);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&t[mode], e0, e1);
  }
  const double mfma = 256.0 * NCW * a.nsteps * 3 * NRP * 16.0, iters = 256.0 * a.nsteps * ((a.NPOS + 15) / 16) * 6.0;
  printf("%-10s%s producers %2d waves x %d tap buffers: consumers %.3f ms (%.1f TFLOP/s executed, %.2f us/step) | producers %.3f ms (%.1f wave-iterations/us/CU, %.2f us/step) | both %.3f ms "
         "(%.2f us/step = %.2f x consumers alone, %.2f x the sum) %s\n", tag, a.prio ? " prio" : "", NPW, NB, t[1], mfma * 2048.0 / (t[1] * 1e-3) / 1e12, 1e3 * t[1] / a.nsteps, t[2],
         iters / 256.0 / (t[2] * 1e3), 1e3 * t[2] / a.nsteps, t[3], 1e3 * t[3] / a.nsteps, t[3] / t[1], t[3] / (t[1] + t[2]), hipGetErrorString(hipGetLastError()));
}

int main() {
  SweepArgs a{};
  a.h = 240; a.w = 320; a.TXI = 35; a.TYI = 18; a.NPOS = 630; a.D = 32; a.PS = 2560; a.nsteps = 96;
  a.lo0 = 1.8f; a.rng = 0.5f;
  // six source views: x / y baselines of 5, 10, 15 cm at f = 250 px (half resolution) and a little rotation
  for (int v = 0; v < 6; ++v) {
    const float bx = (v & 1 ? -1.f : 1.f) * 0.05f * (1 + v / 2) * 250.f, by = (v % 3 == 2 ? 0.02f * 250.f : 0.f), th = 0.01f * (v - 2.5f);
    const float M[12] = {cosf(th), -sinf(th), 160.f * (1 - cosf(th)) + 120.f * sinf(th), bx, sinf(th), cosf(th), 120.f * (1 - cosf(th)) - 160.f * sinf(th), by, 0.f, 0.f, 1.f, 0.f};
    for (int i = 0; i < 12; ++i) a.M[v][i] = M[i];
  }
  for (int c = 0; c < 16; ++c) a.gw[c] = 0.1f + 0.01f * c;
  a.gA1 = 0.7f; a.gB1 = -0.1f; a.gA2 = 0.9f; a.gB2 = 0.05f;
  const size_t nf = (size_t)7 * (a.h + 2) * (a.w + 2) * C;
  std::vector<float> hf(nf);
  unsigned s = 12345;
  for (auto &x : hf) { s = s * 1664525u + 1013904223u; x = (float)(s >> 8) * (1.f / 16777216.f) - 0.5f; }
  float *df, *dout;
  hipMalloc(&df, nf * 4); hipMemcpy(df, hf.data(), nf * 4, hipMemcpyHostToDevice);
  hipMalloc(&dout, (size_t)256 * 64 * 1024 * 16);
  a.feat = df; a.out = dout;
  if (getenv("FS_PRIO")) a.prio = 1;
  run<2, 6>(a, "s2.conv0"); run<4, 2>(a, "s2.conv0"); run<4, 3>(a, "s2.conv0"); run<4, 6>(a, "s2.conv0"); run<6, 3>(a, "s2.conv0"); run<8, 2>(a, "s2.conv0"); run<8, 3>(a, "s2.conv0");
  return 0;
}
