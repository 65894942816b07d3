// Microbenchmark: MFMA rate when the operands of every chunk come from LDS (the conv K loop's structure)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: operands from LDS each chunk; 1: + barrier every 18 chunks; 2: operands constant (register)
__global__ __launch_bounds__(256) void k(float *out, int chunks, int stride) {
  extern __shared__ float4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds4[i] = make_float4(i * 0.001f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  floatx4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  float4 av = lds4[lane], bv[4];
  for (int p = 0; p < 4; ++p) bv[p] = lds4[64 + p * 64 + lane];
  for (int u = 0; u < chunks; ++u) {
    float4 an = av, bn[4];
    for (int p = 0; p < 4; ++p) bn[p] = bv[p];
    if (MODE != 2) {
      const int o = ((u + 1) * stride) & 2047;
      an = lds4[o + lane];
#pragma unroll
      for (int p = 0; p < 4; ++p) bn[p] = lds4[((o + 64 + p * 96) & 2047) + lane];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[p].x, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[p].y, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[p].z, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[p].w, acc[p], 0, 0, 0);
    asm volatile("" ::"v"(an.x), "v"(an.y), "v"(an.z), "v"(an.w));
    av = an;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      asm volatile("" ::"v"(bn[p].x), "v"(bn[p].y), "v"(bn[p].z), "v"(bn[p].w));
      bv[p] = bn[p];
    }
    if (MODE == 1 && (u % 18) == 17) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE>
static void run(const char *name, int blocks_per_cu, size_t lds) {
  float *d;
  const int blocks = 256 * blocks_per_cu, chunks = 18 * 400;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256, lds>>>(d, 180, 7);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256, lds>>>(d, chunks, 7);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * 4 * chunks * 16.0;
  printf("%-44s WG/CU=%d : %.3f ms  %.1f TFLOP/s (%.1f cycles/MFMA/SIMD @2.4GHz)\n", name, blocks_per_cu, ms,
         insts * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (insts / 1024.0));
  hipFree(d);
}
int main() {
  run<2>("operands constant", 1, 65536);
  run<0>("operands from LDS every chunk", 1, 65536);
  run<1>("operands from LDS + barrier / 18 chunks", 1, 65536);
  run<2>("operands constant", 2, 65536);
  run<0>("operands from LDS every chunk", 2, 65536);
  run<1>("operands from LDS + barrier / 18 chunks", 2, 65536);
  return 0;
}
