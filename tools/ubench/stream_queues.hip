// tools/ubench/stream_queues.hip -- which HIP streams of one process share a hardware queue?  Creates N non-blocking streams in order, then for every pair
// launches a one-workgroup spin kernel of ~1 ms on both and times the pair: ~1 ms = they ran side by side (different hardware queues), ~2 ms = one after
// the other (the runtime put both streams on one queue).  The engines' throughput with several windows in flight depends on this mapping
// (profiles/r06_queues_side_stream.txt).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/stream_queues.hip -o /tmp/stream_queues && /tmp/stream_queues [N=10] [touch=1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long ticks, int *sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && ticks < 0) *sink = 1;
}
int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 10, touch = argc > 2 ? atoi(argv[2]) : 1;
  std::vector<hipStream_t> s(N);
  for (int i = 0; i < N; ++i) {
    (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    if (touch) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], 1000LL, (int *)nullptr); (void)hipStreamSynchronize(s[i]); }  // first use right after creation
  }
  const long long ticks = 100000;  // wall_clock64 runs at 100 MHz: 1 ms
  auto pair_ms = [&](int i, int j) {
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], ticks, (int *)nullptr);
    if (j != i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[j], ticks, (int *)nullptr);
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  pair_ms(0, 1);
  printf("streams %d, first use %s; entry = S when the pair serialised (shared queue), . when it overlapped\n    ", N, touch ? "at creation" : "in the test");
  for (int j = 0; j < N; ++j) printf("%2d ", j);
  printf("\n");
  for (int i = 0; i < N; ++i) {
    printf("%2d  ", i);
    for (int j = 0; j < N; ++j) {
      if (j == i) { printf(" - "); continue; }
      const double ms = pair_ms(i, j);
      printf(" %c ", ms > 1.6 ? 'S' : '.');
    }
    printf("\n");
  }
  return 0;
}
