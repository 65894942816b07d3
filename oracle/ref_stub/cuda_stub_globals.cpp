// TEST INFRASTRUCTURE (oracle/): the launch coordinates of oracle/ref_stub/cuda_runtime.h, defined once per shared object.
#include "cuda_runtime.h"
extern "C" {
dim3 threadIdx, blockIdx, blockDim, gridDim;
int cpu_launch_reverse_threads = 0;
unsigned long long cpu_launch_id = 0;
}
