// Semantics check for __builtin_amdgcn_global_load_lds on gfx950: per-lane global source, wave-uniform LDS base +
// lane*16, exec-masked lanes leave LDS untouched, completion via s_waitcnt vmcnt(0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float4 *src, float4 *dst, int n, int mask_mod) {
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n; i += blockDim.x) lds[i] = make_float4(-1.f, -1.f, -1.f, -1.f);
  __syncthreads();
  for (int e0 = wave * 64; e0 < n; e0 += blockDim.x) {
    const int e = e0 + lane;
    // reversed source order to prove the source address is per lane: element e comes from src[n-1-e]
    if (e < n && (mask_mod == 0 || (e % mask_mod) != 0))
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (n - 1 - e)),
                                       (__attribute__((address_space(3))) void *)(lds + e0), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) dst[i] = lds[i];
}
int main() {
  const int n = 1024;
  std::vector<float4> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
  float4 *ds, *dd;
  hipMalloc(&ds, n * 16); hipMalloc(&dd, n * 16);
  hipMemcpy(ds, h.data(), n * 16, hipMemcpyHostToDevice);
  for (int mm : {0, 3}) {
    k<<<1, 256, n * 16>>>(ds, dd, n, mm);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o.data(), dd, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
      const bool masked = mm && (i % mm) == 0;
      const float ex = masked ? -1.f : (float)(n - 1 - i);
      if (o[i].x != ex || (!masked && o[i].w != ex + 0.75f)) { if (bad < 5) printf("  mismatch at %d: got %f %f expected %f\n", i, o[i].x, o[i].w, ex); ++bad; }
    }
    printf("glds test mask_mod=%d: %s (%d bad) err=%d\n", mm, bad ? "FAIL" : "ok", bad, (int)e);
  }
  return 0;
}
