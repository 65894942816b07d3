// rccl_stub.cpp -- a TEST DOUBLE of librccl.so.1 for tests/test_view_shard_gpu.py (TEST INFRASTRUCTURE, never shipped).
//
// RCCL refuses two ranks on one device and the build has a single GPU, so the engine's multi-rank collective path
// (tandem_amd/csrc/dr_mvsnet.hip: ncclReduce of each cost volume to rank 0, ncclBroadcast of the stage depth map back;
// ncclAllReduce under DR_SHARD_ALLREDUCE) could never run with world > 1.  This library implements the five entry points
// the engine binds (plus ncclGetErrorString) for ranks that are PROCESSES SHARING ONE GPU: payloads travel through a
// POSIX shared-memory segment named after the unique id, ranks meet at a sense-reversing barrier in it, sums are taken on
// the host in rank order.  Every call synchronises the caller's stream first and is complete when it returns -- the
// engine's enqueue ORDER, the roots, counts, divisor and in-place semantics are what the test exercises, not overlap.
// The engine loads it through DR_RCCL_LIB.  Prototypes come from the real <rccl/rccl.h>, so a signature drift fails to compile.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {
constexpr size_t kSlot = 32u << 20;  // bytes per rank: the test windows' volumes are a few MB (a container's /dev/shm is small; 157 MB at the headline shape would not fit)
struct Header {
  std::atomic<int> arrived, sense, attached;
  int nranks;
};
}  // namespace
struct ncclComm {
  int rank, nranks;
  Header *hdr;
  char *data;  // nranks slots behind the header
  size_t bytes;
  int local_sense;
  std::string name;
};
namespace {
void barrier(ncclComm *c) {
  c->local_sense ^= 1;
  if (c->hdr->arrived.fetch_add(1) + 1 == c->nranks) {
    c->hdr->arrived.store(0);
    c->hdr->sense.store(c->local_sense);
  } else {
    long spins = 0;
    while (c->hdr->sense.load() != c->local_sense) {
      if (++spins > 2000000000L) { fprintf(stderr, "rccl_stub: barrier timed out on rank %d\n", c->rank); _exit(3); }
      if ((spins & 1023) == 0) usleep(50);
    }
  }
}
size_t size_of(ncclDataType_t t) { return t == ncclFloat ? 4 : (t == ncclDouble ? 8 : (t == ncclInt32 ? 4 : 0)); }
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/drstub_%d_%ld", (int)getpid(), (long)time(nullptr));
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  auto *c = new ncclComm{rank, nranks, nullptr, nullptr, sizeof(Header) + (size_t)nranks * kSlot, 0, std::string(id.internal)};
  int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return ncclSystemError; }
  void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);  // fresh segments read as zero: the header needs no init race
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->hdr = static_cast<Header *>(p);
  c->data = static_cast<char *>(p) + sizeof(Header);
  c->hdr->nranks = nranks;
  c->hdr->attached.fetch_add(1);
  barrier(c);  // like the real call: returns once every rank has joined
  *comm = c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  const bool last = c->hdr->attached.fetch_sub(1) == 1;
  munmap(c->hdr, c->bytes);
  if (last) shm_unlink(c->name.c_str());
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *count) {
  if (!c || !count) return ncclInvalidArgument;
  *count = c->nranks;
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rccl_stub error"; }

ncclResult_t ncclReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, int root, ncclComm_t c, hipStream_t st) {
  const size_t bytes = count * size_of(type);
  if (!c || type != ncclFloat || op != ncclSum || bytes > kSlot || root < 0 || root >= c->nranks) return ncclInvalidArgument;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->data + (size_t)c->rank * kSlot, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  if (c->rank == root) {
    std::vector<float> acc(count);
    memcpy(acc.data(), c->data, bytes);
    for (int r = 1; r < c->nranks; ++r) {
      const float *p = reinterpret_cast<const float *>(c->data + (size_t)r * kSlot);
      for (size_t i = 0; i < count; ++i) acc[i] += p[i];
    }
    if (hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  barrier(c);
  return ncclSuccess;
}
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t c, hipStream_t st) {
  const size_t bytes = count * size_of(type);
  if (!c || !size_of(type) || bytes > kSlot || root < 0 || root >= c->nranks) return ncclInvalidArgument;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (c->rank == root && hipMemcpy(c->data + (size_t)root * kSlot, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  if (c->rank != root && hipMemcpy(recv, c->data + (size_t)root * kSlot, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
  ncclResult_t r = ncclReduce(send, recv, count, type, op, 0, c, st);
  return r != ncclSuccess ? r : ncclBroadcast(recv, recv, count, type, 0, c, st);
}
}  // extern "C"
