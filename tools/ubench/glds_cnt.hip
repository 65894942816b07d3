// Which counters does an LDS-DMA (global_load_lds) hold?  Timeline per wave: t0 | issue N DMAs from cold memory |
// ds_read + s_waitcnt lgkmcnt(0) | t1 | s_waitcnt vmcnt(0) | t2.   If t1-t0 ~ LDS latency, lgkmcnt is independent of
// the DMA and LDS reads can run under outstanding DMAs; if t1-t0 ~ memory latency they cannot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float4 *src, unsigned long long *out, float *sink, int n_dma, int stride) {
  extern __shared__ float4 lds[];
  const int lane = threadIdx.x;
  lds[lane + 4096] = make_float4(1, 2, 3, 4);
  __syncthreads();
  const unsigned ldsb = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)lds);
  unsigned long long t0, t1, t2;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < n_dma; ++i) {
    const unsigned off = ((unsigned)(blockIdx.x * n_dma + i) * stride + lane) * 16u;
    const unsigned m = __builtin_amdgcn_readfirstlane(ldsb + i * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(src), "s"(m) : "memory", "m0");
  }
  float4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ldsb + (lane + 4096) * 16) : "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
  if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t0; }
  if (v.x == 123.f) *sink = v.y;
}
int main() {
  const size_t n = size_t(1) << 26;  // 1 GiB of float4: cold for every block
  float4 *src; unsigned long long *out; float *sink;
  hipMalloc(&src, n * 16); hipMemset(src, 0, n * 16);
  hipMalloc(&out, 1024 * 16); hipMalloc(&sink, 4);
  for (int n_dma : {0, 1, 8, 32}) {
    hipLaunchKernelGGL(k, dim3(256), dim3(64), 160 * 1024 - 64, 0, src, out, sink, n_dma, 4099);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 256; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    printf("n_dma=%2d: t(ds_read+lgkmcnt0) = %.0f ticks, t(vmcnt0) = %.0f ticks (s_memtime, 100 MHz)  err=%d\n", n_dma, a / 256, b / 256, (int)hipGetLastError());
  }
  return 0;
}
