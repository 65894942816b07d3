// glds_probe.hip -- (1) semantics of __builtin_amdgcn_global_load_lds(.., 16, ..) on gfx950: per-lane global source,
// wave-uniform LDS base + lane*16 destination; (2) ds_read_b128 cost of the conv K loop's operand read patterns
// (16 positions x 4 lane groups) for unpadded / padded / swizzled tile layouts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void glb_void;

__global__ void k_sem(const float4 *src, const int *perm, float4 *out) {
  __shared__ float4 buf[256];
  const int tid = threadIdx.x, wave = tid >> 6;
  // lane reads src[perm[tid]] and the hardware drops it at buf[wave*64 + lane]
  __builtin_amdgcn_global_load_lds((const glb_void *)(src + perm[tid]), (lds_void *)(buf + wave * 64), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[tid] = buf[tid];
}

// Operand-read patterns of the conv K loop: lane (j = l & 15, g = l >> 4) reads the float4 of channel group (g % C4) of
// position p0 + j*step (+ tap g / C4).  unit() maps (pos, c4) to a 16-byte LDS slot.  8 independent reads per iteration.
__device__ inline int unit(int mode, int pos, int c4) {
  switch (mode) {
    case 0: return pos * 3 + c4;                                        // CI=8 padded to 12 floats
    case 1: return pos * 2 + c4;                                        // CI=8 unpadded
    case 2: return ((pos ^ ((pos >> 3) & 1)) << 1) | c4;                // CI=8 swizzle A
    case 3: return pos * 5 + c4;                                        // CI=16 padded to 20
    case 4: return pos * 4 + c4;                                        // CI=16 unpadded
    case 5: return (pos << 2) | (c4 ^ ((pos >> 2) & 3));                // CI=16 swizzle A
    case 6: return ((pos ^ ((pos >> 3) & 1)) << 2) | (c4 ^ ((pos >> 1) & 3));  // CI=16 swizzle B
    case 7: return ((pos ^ ((pos >> 2) & 1)) << 2) | (c4 ^ ((pos >> 1) & 1) ^ (((pos >> 3) & 1) << 1));  // CI=16 swizzle C
    case 8: return pos;                                                 // CI=4 unpadded
    default: return pos * 2;                                            // CI=4 padded to 8
  }
}
__global__ void k_read(float *out, int mode, int c4n, int step, int iters, long long *cyc) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  int off[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) off[u] = (unit(mode, 5 + j * step + (g / c4n) + u * 7, g % c4n) * 4) & 16380;
  float4 acc = make_float4(0, 0, 0, 0);
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(lds + off[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    asm volatile("" : "+v"(off[0]), "+v"(off[1]), "+v"(off[2]), "+v"(off[3]), "+v"(off[4]), "+v"(off[5]), "+v"(off[6]), "+v"(off[7]));
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  float4 *src, *out; int *perm;
  CK(hipMalloc(&src, 1024 * 16)); CK(hipMalloc(&out, 256 * 16)); CK(hipMalloc(&perm, 256 * 4));
  std::vector<float4> h(1024); for (int i = 0; i < 1024; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
  std::vector<int> p(256); for (int i = 0; i < 256; ++i) p[i] = (i * 37 + 11) % 1024;
  CK(hipMemcpy(src, h.data(), 1024 * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(perm, p.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sem, dim3(1), dim3(256), 0, 0, src, perm, out);
  std::vector<float4> o(256); CK(hipMemcpy(o.data(), out, 256 * 16, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; ++i) if (o[i].x != (float)p[i] || o[i].w != p[i] + 0.75f) ++bad;
  printf("glds semantics: %d / 256 lanes wrong\n", bad);
  float *fo; long long *cy; CK(hipMalloc(&fo, 512 * 4 * 4)); CK(hipMalloc(&cy, 64));
  struct { const char *name; int mode, c4n; } cases[] = {{"CI8 padded 12", 0, 2}, {"CI8 unpadded", 1, 2}, {"CI8 swizzle A", 2, 2},
      {"CI16 padded 20", 3, 4}, {"CI16 unpadded", 4, 4}, {"CI16 swizzle A", 5, 4}, {"CI16 swizzle B", 6, 4}, {"CI16 swizzle C", 7, 4},
      {"CI4 unpadded", 8, 1}, {"CI4 padded 8", 9, 1}};
  for (auto &c : cases)
    for (int step : {1, 2}) {
      printf("%-16s stride %d:", c.name, step);
      for (int waves : {1, 4, 8}) {
        hipLaunchKernelGGL(k_read, dim3(1), dim3(64 * waves), 65536, 0, fo, c.mode, c.c4n, step, 200, cy);
        long long t; CK(hipMemcpy(&t, cy, 8, hipMemcpyDeviceToHost));
        printf("  %dw %.1f", waves, (double)t / (200 * 8));
      }
      printf("   cycles per wave ds_read_b128\n");
    }
  return 0;
}
