// Stand-in for the few RCCL entry points the solver binds (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllGather; ncclAllReduce / ncclGroupStart/End are kept for older builds), for
// TESTS ONLY: the ranks are processes -- or threads of one process -- that share ONE GPU, which
// real RCCL refuses, and exchange through a POSIX shared-memory segment.  Collectives are executed synchronously on the host (stream sync, D2H,
// barrier, reduce in rank order, H2D): slow, but the solver's communicator code path (grouped
// sum/min/max all-reduces on adjacent segments, all-gathers of record chunks, counts, offsets)
// runs exactly as with the real library.  Selected with LBFGSB_RCCL_LIBRARY.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr size_t SLOT = (size_t)128 << 20;  // bytes per rank (only touched pages are ever backed)
struct Shared {
  std::atomic<int> arrived;
  std::atomic<int> generation;
};
struct Comm {
  int rank, nranks;
  char name[64];
  size_t bytes;
  Shared *sh;
  char *slots;
};
void barrier(Comm *c) {
  const int gen = c->sh->generation.load();
  if (c->sh->arrived.fetch_add(1) + 1 == c->nranks) {
    c->sh->arrived.store(0);
    c->sh->generation.fetch_add(1);
  } else {
    while (c->sh->generation.load() == gen) sched_yield();
  }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0, sizeof *id);
  std::snprintf(id->internal, sizeof id->internal, "/lbfgsb_fake_rccl_%d_%ld", (int)getpid(), random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  Comm *c = new Comm;
  c->rank = rank, c->nranks = nranks;
  std::snprintf(c->name, sizeof c->name, "%s", id.internal);
  c->bytes = 4096 + (size_t)nranks * SLOT;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError;
  void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->sh = static_cast<Shared *>(p);  // (a fresh segment is zero-filled: counters start at 0)
  c->slots = static_cast<char *>(p) + 4096;
  *out = reinterpret_cast<ncclComm_t>(c);
  barrier(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  barrier(c);
  munmap(c->sh, c->bytes);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (dt != ncclDouble || count * 8 > SLOT) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slots + (size_t)c->rank * SLOT, send, count * 8, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  barrier(c);
  std::vector<double> acc(count);
  for (size_t k = 0; k < count; ++k) {
    double v = reinterpret_cast<double *>(c->slots)[k];
    for (int r = 1; r < c->nranks; ++r) {
      const double w = reinterpret_cast<double *>(c->slots + (size_t)r * SLOT)[k];
      v = op == ncclSum ? v + w : op == ncclMin ? (w < v ? w : v) : op == ncclMax ? (w > v ? w : v) : v;
    }
    acc[k] = v;
  }
  barrier(c);  // everyone has read the slots before they are reused
  if (hipMemcpy(recv, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess)
    return ncclUnhandledCudaError;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (dt != ncclDouble || count * 8 > SLOT) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slots + (size_t)c->rank * SLOT, send, count * 8, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  barrier(c);
  for (int r = 0; r < c->nranks; ++r)
    if (hipMemcpy(static_cast<char *>(recv) + (size_t)r * count * 8, c->slots + (size_t)r * SLOT, count * 8,
                  hipMemcpyHostToDevice) != hipSuccess)
      return ncclUnhandledCudaError;
  barrier(c);
  return ncclSuccess;
}

}  // extern "C"
