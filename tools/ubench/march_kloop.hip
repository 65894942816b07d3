// Microbenchmark: the K loop of k_conv_m (march_kloop, conv_march.h) on its own -- LDS-fed fp32 MFMA rate as a function of
// the prefetch depth, the position tiles per wave, the waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 march_kloop.hip
#include <cstdio>
#include "../../tandem_amd/csrc/conv_mfma.h"
namespace dr { std::string &last_error_slot() { static std::string s; return s; } }
using namespace dr;

template <int NUP, int CT, int PT, int DEPTH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float *out, int nsec, int PS, int txi) {
  extern __shared__ float4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, g = lane >> 4;
  for (int i = tid; i < 3 * PS + 3 * NUP * CT * 64; i += 64 * WAVES) lds4[i] = make_float4(i * 1e-4f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  int sw[NUP][PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int u = 0; u < NUP; ++u) {
      const int bpos = ((wave % 8) * PT + pt) * txi + j * 2;  // XPAIR addressing: positions two pixels apart
      sw[u][pt] = conv_a_unit<16>(bpos + (u / 4) * txi + (u % 4), g);
      asm volatile("" : "+v"(sw[u][pt]));
    }
  floatx4 acc[CT][PT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = floatx4{0.f, 0.f, 0.f, 0.f};
  const float4 *wl = lds4 + 3 * PS + lane;
  for (int s = 0; s < nsec; ++s) march_kloop<NUP, CT, PT, DEPTH>(lds4 + (s % 3) * PS, wl + (s % 3) * NUP * CT * 64, sw, acc);
  float r = 0;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) r += acc[ct][pt][0] + acc[ct][pt][1] + acc[ct][pt][2] + acc[ct][pt][3];
  out[blockIdx.x * 64 * WAVES + tid] = r;
}

template <int NUP, int CT, int PT, int DEPTH, int WAVES>
static void run(int wg_per_cu) {
  const int PS = 2560, txi = 34, nsec = 3 * 200;
  const size_t lds = ((size_t)3 * PS + 3 * NUP * CT * 64) * 16;
  float *d;
  const int blocks = 256 * wg_per_cu;
  hipMalloc(&d, (size_t)blocks * 64 * WAVES * 4);
  auto kern = k<NUP, CT, PT, DEPTH, WAVES>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 64 * WAVES, lds>>>(d, 30, PS, txi);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, 64 * WAVES, lds>>>(d, nsec, PS, txi);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * WAVES * nsec * NUP * 4.0 * CT * PT;
  printf("NUP %2d CT %d PT %d depth %d waves/WG %2d WG/CU %d (LDS %3zu KB): %.3f ms %6.1f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz (%s)\n", NUP, CT, PT, DEPTH, WAVES, wg_per_cu, lds >> 10, ms,
         mfma * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (mfma / 1024.0), hipGetErrorString(hipGetLastError()));
  hipFree(d);
}
int main() {
  run<12, 1, 2, 1, 8>(1); run<12, 1, 2, 2, 8>(1); run<12, 1, 2, 3, 8>(1);
  run<12, 1, 2, 1, 4>(1); run<12, 1, 2, 2, 4>(1);
  run<12, 1, 4, 1, 8>(1); run<12, 1, 4, 2, 8>(1);
  run<12, 1, 2, 2, 12>(1); run<12, 1, 2, 2, 16>(1); run<12, 1, 1, 2, 16>(1);
  run<9, 2, 2, 1, 8>(1); run<9, 2, 2, 2, 8>(1);
  return 0;
}
