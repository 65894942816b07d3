// TEST INFRASTRUCTURE (oracle/): a CPU stand-in for the few pieces of the CUDA runtime that the reference's
// dr_fusion / cuda_coarse_tracker sources use, so that `oracle/Makefile.ref` can compile those sources WHERE THEY LIE under
// /root/reference with g++ and run their kernels serially on the host (oracle/_ref/*.so).  Nothing here is product code
// and nothing here comes from the reference: it is the glue that lets the reference's own arithmetic execute in this
// container, so that oracle/tsdf_oracle.c (the restatement that travels to the GPU box) can be pinned against it.
//
// Execution model: a kernel launch `k<<<grid, block, shmem, stream>>>(args...)` is rewritten by the build recipe (sed, piped
// straight into the compiler -- no copy of the source is kept) to `cpu_launch(grid, block, k, args...)`, which calls the
// kernel body once per (block, thread) in ascending order with blockIdx/threadIdx/blockDim/gridDim set.  Atomics are the
// plain serial operations; streams/events are no-ops (everything is synchronous); managed/pinned/device memory is calloc.
// Consequences of the serial schedule, all of them *legal* schedules of the CUDA program: the try-lock of
// HashTable::AllocateBlock never sees contention (no dropped inserts), heap slots are consumed in thread order.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstddef>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
struct uchar3 { unsigned char x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return uint3{x, y, z}; }
static inline uchar3 make_uchar3(unsigned char x, unsigned char y, unsigned char z) { return uchar3{x, y, z}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }

// one definition per shared object: every translation unit sees the same launch coordinates
extern "C" {
extern dim3 threadIdx, blockIdx, blockDim, gridDim;
}

// `rev` runs the threads of a block in DESCENDING order (used for kernels whose thread 0 publishes a block reduction)
extern "C" int cpu_launch_reverse_threads;
extern "C" unsigned long long cpu_launch_id;
template <class... P, class... A>
static inline void cpu_launch(dim3 grid, dim3 block, void (*kernel)(P...), A&&... args) {
  ++cpu_launch_id;
  gridDim = grid;
  blockDim = block;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned t = 0; t < block.x; ++t) {
              unsigned tx = cpu_launch_reverse_threads ? block.x - 1 - t : t;
              threadIdx = dim3(tx, ty, tz);
              kernel(args...);
            }
      }
}

// ---- atomics (serial) ----
template <class T> static inline T atomicCAS(T* a, T cmp, T val) { T old = *a; if (old == cmp) *a = val; return old; }
template <class T, class U> static inline T atomicAdd(T* a, U v) { T old = *a; *a = (T)(old + (T)v); return old; }
template <class T, class U> static inline T atomicSub(T* a, U v) { T old = *a; *a = (T)(old - (T)v); return old; }
template <class T, class U> static inline T atomicMin(T* a, U v) { T old = *a; if ((T)v < old) *a = (T)v; return old; }
template <class T, class U> static inline T atomicMax(T* a, U v) { T old = *a; if ((T)v > old) *a = (T)v; return old; }
template <class T, class U> static inline T atomicExch(T* a, U v) { T old = *a; *a = (T)v; return old; }
static inline void __syncthreads() {}
static inline void __threadfence() {}

// ---- runtime API subset ----
typedef int cudaError_t;
enum { cudaSuccess = 0 };
typedef void* cudaStream_t;
typedef void* cudaEvent_t;
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice, cudaMemcpyDefault };
enum { cudaEventBlockingSync = 1, cudaEventDisableTiming = 2, cudaStreamNonBlocking = 1, cudaStreamDefault = 0 };
static inline const char* cudaGetErrorString(cudaError_t) { return "cpu-stub"; }
template <class T> static inline cudaError_t cudaMallocManaged(T** p, size_t n, unsigned = 1) { *p = (T*)calloc(1, n ? n : 1); return *p ? 0 : 2; }
template <class T> static inline cudaError_t cudaMalloc(T** p, size_t n) { *p = (T*)calloc(1, n ? n : 1); return *p ? 0 : 2; }
template <class T> static inline cudaError_t cudaMallocHost(T** p, size_t n) { *p = (T*)calloc(1, n ? n : 1); return *p ? 0 : 2; }
template <class T> static inline cudaError_t cudaHostAlloc(T** p, size_t n, unsigned) { *p = (T*)calloc(1, n ? n : 1); return *p ? 0 : 2; }
static inline cudaError_t cudaFree(void* p) { free(p); return 0; }
static inline cudaError_t cudaFreeHost(void* p) { free(p); return 0; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return 0; }
static inline cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind, cudaStream_t = nullptr) { memcpy(d, s, n); return 0; }
static inline cudaError_t cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline cudaError_t cudaMemsetAsync(void* d, int v, size_t n, cudaStream_t = nullptr) { memset(d, v, n); return 0; }
static inline cudaError_t cudaDeviceSynchronize() { return 0; }
static inline cudaError_t cudaPeekAtLastError() { return 0; }
static inline cudaError_t cudaGetLastError() { return 0; }
static inline cudaError_t cudaSetDevice(int) { return 0; }
static inline cudaError_t cudaDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return 0; }
static inline cudaError_t cudaStreamCreate(cudaStream_t* s) { *s = nullptr; return 0; }
static inline cudaError_t cudaStreamCreateWithFlags(cudaStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline cudaError_t cudaStreamCreateWithPriority(cudaStream_t* s, unsigned, int) { *s = nullptr; return 0; }
static inline cudaError_t cudaStreamDestroy(cudaStream_t) { return 0; }
static inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return 0; }
static inline cudaError_t cudaStreamWaitEvent(cudaStream_t, cudaEvent_t, unsigned) { return 0; }
static inline cudaError_t cudaEventCreate(cudaEvent_t* e) { *e = nullptr; return 0; }
static inline cudaError_t cudaEventCreateWithFlags(cudaEvent_t* e, unsigned) { *e = nullptr; return 0; }
static inline cudaError_t cudaEventDestroy(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t = nullptr) { return 0; }
static inline cudaError_t cudaEventSynchronize(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventQuery(cudaEvent_t) { return 0; }
static inline cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t, cudaEvent_t) { *ms = 0.0f; return 0; }
// CUDA puts the <math.h> classification functions in the global namespace for device code
using std::isfinite;
using std::abs;      // CUDA overloads ::abs for float/double; without this g++ would pick int abs(int)
using std::isnan;
using std::isinf;
