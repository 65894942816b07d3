// TEST INFRASTRUCTURE (oracle/ref_stub): cub::BlockReduce for the serial host launch of ref_stub/cuda_runtime.h.
// The launch runs the threads of a block one after the other, in DESCENDING threadIdx.x when cpu_launch_reverse_threads
// is set; Sum() then returns the running total of the k-th reduction of the block, so that thread 0 -- the one thread
// whose return value cub defines, and which the reference lets publish it with atomicAdd -- runs last and sees the sum
// of all threads.  The order of the additions (thread 127 ... 0) is one of the orders the real block reduction may use
// (cub documents it as unspecified for floating point).
#pragma once
#include "cuda_runtime.h"
extern "C" unsigned long long cpu_launch_id;
namespace cub {
enum BlockReduceAlgorithm { BLOCK_REDUCE_RAKING_COMMUTATIVE_ONLY, BLOCK_REDUCE_RAKING, BLOCK_REDUCE_WARP_REDUCTIONS };
template <class T, int TPB, BlockReduceAlgorithm A = BLOCK_REDUCE_WARP_REDUCTIONS>
class BlockReduce {
 public:
  struct TempStorage {
    T acc[64];
    int call;
    unsigned tid, blk;
    unsigned long long launch;
  };
  explicit BlockReduce(TempStorage& s) : s_(s) {}
  T Sum(T x) {
    if (!cpu_launch_reverse_threads) abort();  // thread 0 must run last
    if (s_.launch != cpu_launch_id || s_.blk != blockIdx.x || s_.tid != threadIdx.x) {
      s_.launch = cpu_launch_id; s_.blk = blockIdx.x; s_.tid = threadIdx.x; s_.call = 0;
    }
    if (s_.call >= 64) abort();
    if (threadIdx.x == blockDim.x - 1) s_.acc[s_.call] = T(0);
    s_.acc[s_.call] += x;
    return s_.acc[s_.call++];
  }
 private:
  TempStorage& s_;
};
}  // namespace cub
