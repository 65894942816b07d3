// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2_f32 from 1..N waves per SIMD (calibrates the roofline peak)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float *out, int iters, float a, float b) {
  floatx4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float *out, int iters, float a, float b) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K>
static void run(const char *name, K kern, int waves_per_simd, int nacc, double flop_per_inst) {
  float *d;
  const int blocks = 256 * waves_per_simd, threads = 256, iters = 4000;
  hipMalloc(&d, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, threads>>>(d, 100, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, threads>>>(d, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * (threads / 64) * iters * 4.0 * nacc;
  printf("%-28s waves/SIMD=%d acc=%d : %.3f ms  %.1f TFLOP/s  (%.1f cycles/inst/SIMD @2.4GHz)\n", name, waves_per_simd, nacc, ms,
         insts * flop_per_inst / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (insts / 1024.0));
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) {
    run("mfma_f32_16x16x4_f32", k16<4>, w, 4, 2048.0);
    run("mfma_f32_16x16x4_f32", k16<2>, w, 2, 2048.0);
    run("mfma_f32_16x16x4_f32", k16<1>, w, 1, 2048.0);
    run("mfma_f32_32x32x2_f32", k32<4>, w, 4, 4096.0);
    run("mfma_f32_32x32x2_f32", k32<1>, w, 1, 4096.0);
  }
  return 0;
}
