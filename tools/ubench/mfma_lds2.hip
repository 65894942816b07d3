// Microbenchmark 2: what an LDS operand read costs the fp32 MFMA stream.  Same structure as mfma_lds.hip (one chunk ahead),
// varying (a) the MFMA shape (16x16x4: 16 MFMAs of 32 cycles per 5 reads; 32x32x2: 16 MFMAs of 64 cycles per 5 reads),
// (b) ping-pong operand registers (no copies at the end of the iteration), (c) the number of reads per chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NB>  // 16x16x4, ping-pong registers, NB position tiles per wave (NB B reads + 1 A read per 4*NB MFMAs)
__global__ __launch_bounds__(256) void k16(float *out, int chunks, int stride) {
  extern __shared__ float4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds4[i] = make_float4(i * 0.001f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  floatx4 acc[NB];
  for (int i = 0; i < NB; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  float4 a0 = lds4[lane], b0[NB], a1, b1[NB];
  for (int p = 0; p < NB; ++p) b0[p] = lds4[64 + p * 64 + lane];
  for (int u = 0; u < chunks; u += 2) {
    int o = ((u + 1) * stride) & 2047;
    a1 = lds4[o + lane];
#pragma unroll
    for (int p = 0; p < NB; ++p) b1[p] = lds4[((o + 64 + p * 96) & 2047) + lane];
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0[p].x, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0[p].y, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0[p].z, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0[p].w, acc[p], 0, 0, 0);
    asm volatile("" ::"v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w));
#pragma unroll
    for (int p = 0; p < NB; ++p) asm volatile("" ::"v"(b1[p].x), "v"(b1[p].y), "v"(b1[p].z), "v"(b1[p].w));
    o = ((u + 2) * stride) & 2047;
    a0 = lds4[o + lane];
#pragma unroll
    for (int p = 0; p < NB; ++p) b0[p] = lds4[((o + 64 + p * 96) & 2047) + lane];
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1[p].x, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1[p].y, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1[p].z, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1[p].w, acc[p], 0, 0, 0);
    asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w));
#pragma unroll
    for (int p = 0; p < NB; ++p) asm volatile("" ::"v"(b0[p].x), "v"(b0[p].y), "v"(b0[p].z), "v"(b0[p].w));
  }
  float s = 0;
  for (int i = 0; i < NB; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + tid] = s;
}

// 16x16x4 with the conv kernel's B-operand addressing: lane (j = l & 15, g = l >> 4) reads 16 B at
//   float offset  base[p] + tap[u] ,  base = j * STRIDE + (4 g) % CI,  tap selected by (4 g) / CI   (CI = 8, row stride CI + 4 = 12)
// STRIDE = 12 (stride-1 layers) or 24 (XPAIR layers: positions are 2 pixels apart); A (weights) stays linear.
template <int STRIDE>
__global__ __launch_bounds__(256) void k16pat(float *out, int chunks, int stride) {
  extern __shared__ float4 lds4[];
  const float *lds = reinterpret_cast<const float *>(lds4);
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  for (int i = tid; i < 4096; i += 256) lds4[i] = make_float4(i * 0.001f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  floatx4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  int base[4];
  for (int p = 0; p < 4; ++p) base[p] = (p * 16 + j) * STRIDE + (4 * g) % 8 + ((4 * g) / 8) * 12;
  float4 a0 = lds4[lane], b0[4], a1, b1[4];
  for (int p = 0; p < 4; ++p) b0[p] = *reinterpret_cast<const float4 *>(lds + base[p]);
  for (int u = 0; u < chunks; u += 2) {
    int o = (((u + 1) * stride) & 255) * 12;
    a1 = lds4[((u + 1) & 15) * 64 + lane];
#pragma unroll
    for (int p = 0; p < 4; ++p) b1[p] = *reinterpret_cast<const float4 *>(lds + base[p] + o);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0[p].x, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0[p].y, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0[p].z, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0[p].w, acc[p], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::"v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w));
#pragma unroll
    for (int p = 0; p < 4; ++p) asm volatile("" ::"v"(b1[p].x), "v"(b1[p].y), "v"(b1[p].z), "v"(b1[p].w));
    o = (((u + 2) * stride) & 255) * 12;
    a0 = lds4[((u + 2) & 15) * 64 + lane];
#pragma unroll
    for (int p = 0; p < 4; ++p) b0[p] = *reinterpret_cast<const float4 *>(lds + base[p] + o);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1[p].x, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1[p].y, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1[p].z, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1[p].w, acc[p], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w));
#pragma unroll
    for (int p = 0; p < 4; ++p) asm volatile("" ::"v"(b0[p].x), "v"(b0[p].y), "v"(b0[p].z), "v"(b0[p].w));
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NB>  // 32x32x2: NB position tiles (32 positions each) per wave, 32 rows; 1 A + NB B reads per 4*NB MFMAs of 64 cycles
__global__ __launch_bounds__(256) void k32(float *out, int chunks, int stride) {
  extern __shared__ float4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds4[i] = make_float4(i * 0.001f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  floatx16 acc[NB];
  for (int i = 0; i < NB; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float4 a0 = lds4[lane], b0[NB], a1, b1[NB];
  for (int p = 0; p < NB; ++p) b0[p] = lds4[64 + p * 64 + lane];
  for (int u = 0; u < chunks; u += 2) {
    int o = ((u + 1) * stride) & 2047;
    a1 = lds4[o + lane];
#pragma unroll
    for (int p = 0; p < NB; ++p) b1[p] = lds4[((o + 64 + p * 96) & 2047) + lane];
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0[p].x, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0[p].y, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0[p].z, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0[p].w, acc[p], 0, 0, 0);
    asm volatile("" ::"v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w));
#pragma unroll
    for (int p = 0; p < NB; ++p) asm volatile("" ::"v"(b1[p].x), "v"(b1[p].y), "v"(b1[p].z), "v"(b1[p].w));
    o = ((u + 2) * stride) & 2047;
    a0 = lds4[o + lane];
#pragma unroll
    for (int p = 0; p < NB; ++p) b0[p] = lds4[((o + 64 + p * 96) & 2047) + lane];
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1[p].x, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1[p].y, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1[p].z, acc[p], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1[p].w, acc[p], 0, 0, 0);
    asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w));
#pragma unroll
    for (int p = 0; p < NB; ++p) asm volatile("" ::"v"(b0[p].x), "v"(b0[p].y), "v"(b0[p].z), "v"(b0[p].w));
  }
  float s = 0;
  for (int i = 0; i < NB; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + tid] = s;
}

template <class K>
static void run(const char *name, K kern, int blocks_per_cu, int mfma_per_chunk, double flop_per_mfma, double cyc_per_mfma) {
  float *d;
  const int blocks = 256 * blocks_per_cu, chunks = 18 * 400;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 256, 65536>>>(d, 180, 7);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, 256, 65536>>>(d, chunks, 7);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * 4 * chunks * mfma_per_chunk;
  printf("%-52s WG/CU=%d : %.3f ms  %6.1f TFLOP/s (%.1f cycles/MFMA/SIMD, ideal %.0f)\n", name, blocks_per_cu, ms,
         insts * flop_per_mfma / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (insts / 1024.0), cyc_per_mfma);
  hipFree(d);
}
int main() {
  for (int w = 1; w <= 2; ++w) {
    run("16x16x4 ping-pong, 1 A + 4 B reads / 16 MFMA", k16<4>, w, 16, 2048.0, 32);
    run("16x16x4 conv addressing, lane stride 12 floats", k16pat<12>, w, 16, 2048.0, 32);
    run("16x16x4 conv addressing, lane stride 24 floats", k16pat<24>, w, 16, 2048.0, 32);
    run("16x16x4 conv addressing, lane stride 20 floats", k16pat<20>, w, 16, 2048.0, 32);
    run("16x16x4 ping-pong, 1 A + 8 B reads / 32 MFMA", k16<8>, w, 32, 2048.0, 32);
    run("16x16x4 ping-pong, 1 A + 2 B reads /  8 MFMA", k16<2>, w, 8, 2048.0, 32);
    run("32x32x2 ping-pong, 1 A + 2 B reads /  8 MFMA", k32<2>, w, 8, 4096.0, 64);
    run("32x32x2 ping-pong, 1 A + 4 B reads / 16 MFMA", k32<4>, w, 16, 4096.0, 64);
  }
  return 0;
}
