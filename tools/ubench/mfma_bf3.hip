// Microbenchmark for the bf16 x 3 K loop (csrc/conv_bf3.h): what limits it -- MFMA issue (3 x 16 cycles per 32-wide chunk, row tile and
// position tile) or the LDS operand reads (per chunk: 2 weight fragments per row tile + 2 input fragments per position tile, 16 B each)?
// Same structure as mfma_lds2.hip (ping-pong operand sets, one chunk ahead), the conv kernel's B addressing (lane (j, g): record
// (position j) * RB + a tap offset, hi at +0 and lo at +2 CI bytes), A fragments linear.  Reported per configuration: time, the
// algorithmic fp32-equivalent TFLOP/s (2 * 16 * 16 * 32 per chunk and tile pair -- what the fp32 kernel would have to do), cycles
// per chunk and SIMD against the MFMA floor (48 * CT * PT) and the LDS floor (bytes / 128 B/clk/CU, shared by the CU's waves).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf3.hip -o tools/ubench/mfma_bf3 && tools/ubench/mfma_bf3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CI, int CT, int PT, int SX>  // SX: positions a lane's neighbour is apart (1: stride-1 layers, 2: XPAIR)
__global__ __launch_bounds__(256) void kb(float *out, int chunks) {
  extern __shared__ float4 lds4[];
  const char *ldsb = reinterpret_cast<const char *>(lds4);
  constexpr int RB = (CI + 4) * 4, LO = 2 * CI;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  for (int i = tid; i < 4096; i += 256) lds4[i] = make_float4(1.f, 1.f, 1.f, 1.f);  // (bit patterns of small bf16 pairs; values do not matter)
  __syncthreads();
  floatx4 acc[CT][PT];
  for (int c = 0; c < CT; ++c) for (int p = 0; p < PT; ++p) acc[c][p] = floatx4{0.f, 0.f, 0.f, 0.f};
  int baseb[PT];
  for (int p = 0; p < PT; ++p) baseb[p] = (((wave * PT + p) * 16 + j) * SX % 256) * RB + ((8 * g) % CI) * 2 + ((8 * g) / CI) * RB;
  const float4 *wp = lds4 + 2048 + lane;  // weight fragments: the upper half of the 64 KB
  float4 ah0[CT], al0[CT], bh0[PT], bl0[PT], ah1[CT], al1[CT], bh1[PT], bl1[PT];
  auto load = [&](int u, float4 (&ah)[CT], float4 (&al)[CT], float4 (&bh)[PT], float4 (&bl)[PT]) {
    const int o = ((u * 7) & 63) * RB;
#pragma unroll
    for (int c = 0; c < CT; ++c) { ah[c] = wp[(((u * CT + c) * 2) & 31) * 64]; al[c] = wp[(((u * CT + c) * 2 + 1) & 31) * 64]; }
#pragma unroll
    for (int p = 0; p < PT; ++p) { bh[p] = *reinterpret_cast<const float4 *>(ldsb + baseb[p] + o); bl[p] = *reinterpret_cast<const float4 *>(ldsb + baseb[p] + o + LO); }
  };
  auto mfma = [&](const float4 (&ah)[CT], const float4 (&al)[CT], const float4 (&bh)[PT], const float4 (&bl)[PT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int p = 0; p < PT; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[c]), __builtin_bit_cast(bf16x8, bh[p]), acc[c][p], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int p = 0; p < PT; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[c]), __builtin_bit_cast(bf16x8, bl[p]), acc[c][p], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int p = 0; p < PT; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[c]), __builtin_bit_cast(bf16x8, bh[p]), acc[c][p], 0, 0, 0);
  };
  auto anchor = [&](const float4 (&ah)[CT], const float4 (&al)[CT], const float4 (&bh)[PT], const float4 (&bl)[PT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c) { asm volatile("" ::"v"(ah[c].x), "v"(ah[c].y), "v"(ah[c].z), "v"(ah[c].w)); asm volatile("" ::"v"(al[c].x), "v"(al[c].y), "v"(al[c].z), "v"(al[c].w)); }
#pragma unroll
    for (int p = 0; p < PT; ++p) { asm volatile("" ::"v"(bh[p].x), "v"(bh[p].y), "v"(bh[p].z), "v"(bh[p].w)); asm volatile("" ::"v"(bl[p].x), "v"(bl[p].y), "v"(bl[p].z), "v"(bl[p].w)); }
  };
  load(0, ah0, al0, bh0, bl0);
  for (int u = 0; u < chunks; u += 2) {
    load(u + 1, ah1, al1, bh1, bl1);
    __builtin_amdgcn_sched_barrier(0);
    mfma(ah0, al0, bh0, bl0);
    __builtin_amdgcn_sched_barrier(0);
    anchor(ah1, al1, bh1, bl1);
    load(u + 2, ah0, al0, bh0, bl0);
    __builtin_amdgcn_sched_barrier(0);
    mfma(ah1, al1, bh1, bl1);
    __builtin_amdgcn_sched_barrier(0);
    anchor(ah0, al0, bh0, bl0);
  }
  float s = 0;
  for (int c = 0; c < CT; ++c) for (int p = 0; p < PT; ++p) s += acc[c][p][0] + acc[c][p][1] + acc[c][p][2] + acc[c][p][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <class K>
static void run(const char *name, K kern, int blocks_per_cu, int ct, int pt) {
  float *d;
  const int blocks = 256 * blocks_per_cu, chunks = 9 * 800;
  (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kern<<<blocks, 256, 65536>>>(d, 90);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<blocks, 256, 65536>>>(d, chunks);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double wave_chunks = (double)blocks * 4 * chunks;                // chunks executed by all waves
  const double cyc = ms * 1e-3 * 2.4e9 / (wave_chunks / 1024.0);         // cycles per chunk and SIMD (1024 SIMDs)
  const double lds_floor = (2.0 * ct + 2.0 * pt) * 1024.0 / 128.0 * 4.0;  // a wave reads (2 CT + 2 PT) KB per chunk at the CU's 128 B/clk, which its 4 SIMDs share
  printf("%-34s WG/CU=%d : %.3f ms  %6.1f fp32-equivalent TFLOP/s  %.0f cycles/chunk/SIMD (MFMA floor %d, LDS floor %.0f)\n", name, blocks_per_cu, ms,
         wave_chunks * ct * pt * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12, cyc, 48 * ct * pt, lds_floor);
  (void)hipFree(d);
}
int main() {
  for (int w = 1; w <= 2; ++w) {
    run("CI=16 CT=1 PT=1 stride 1", kb<16, 1, 1, 1>, w, 1, 1);
    run("CI=16 CT=1 PT=4 stride 1", kb<16, 1, 4, 1>, w, 1, 4);
    run("CI=16 CT=1 PT=4 XPAIR", kb<16, 1, 4, 2>, w, 1, 4);
    run("CI=8  CT=1 PT=4 stride 1", kb<8, 1, 4, 1>, w, 1, 4);
    run("CI=8  CT=1 PT=4 XPAIR", kb<8, 1, 4, 2>, w, 1, 4);
    run("CI=16 CT=2 PT=4 stride 1", kb<16, 2, 4, 1>, w, 2, 4);
    run("CI=16 CT=4 PT=4 stride 1", kb<16, 4, 4, 1>, w, 4, 4);
    run("CI=16 CT=2 PT=1 stride 1", kb<16, 2, 1, 1>, w, 2, 1);
    run("CI=16 CT=4 PT=1 stride 1", kb<16, 4, 1, 1>, w, 4, 1);
  }
  return 0;
}
