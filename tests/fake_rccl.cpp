// Stand-in for the few RCCL entry points the solver binds (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllGather, ncclCommCount, ncclCommUserRank), for TESTS ONLY: the ranks are
// processes -- or threads of one process -- that share ONE GPU, which real RCCL refuses, and exchange
// through a POSIX shared-memory segment.  Built with hipcc (it launches one tiny kernel).
//
// It fails the way RCCL fails:
//  * ASYNCHRONOUS on the caller's stream.  ncclAllGather only ENQUEUES -- a D2H copy of the send buffer
//    into pinned staging, a one-wave kernel that waits for this rank's exchange to finish, an H2D copy
//    of the gathered result -- and returns before any data has moved; the exchange itself (shared-memory
//    slots, barriers) runs on a worker thread of the communicator.  A caller that reuses a host buffer,
//    or reads a result, before the STREAM got there sees stale data here exactly as with the real library.
//    (Round 3's stand-in synchronised the stream inside every call and hid that class of bug.)
//  * COUNT / DATATYPE CHECK.  Every rank publishes the element count and datatype of each collective;
//    a mismatch -- which hangs or corrupts with real RCCL -- poisons the result with NaN, prints the two
//    counts, and makes every later call on the communicator return ncclInvalidArgument.
//  * In-order matching: the k-th collective of a rank meets the k-th of every other rank.
// LBFGSB_FAKE_RCCL_SYNC=1 selects the synchronous form (stream sync, copies and exchange inside the call;
// the count check stays): for ranks that are THREADS of one process.  There a device-wide synchronising
// runtime call of one rank (hipFree, hipMalloc growth of a buffer) waits for the OTHER ranks' streams too,
// and a stream-side wait for a collective that needs the blocked rank's next contribution never ends -- the
// same rule real RCCL documents (one process per GPU is the supported shape, and the one bench.py uses).
// Selected with LBFGSB_RCCL_LIBRARY.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {
constexpr size_t SLOT = (size_t)128 << 20;  // bytes per rank (only touched pages are ever backed)
constexpr int MAXR = 64;
struct Shared {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  // what every rank says about the collective it is in (written before the first barrier of the op)
  struct {
    unsigned long long count;
    int dtype;
    int pad;
  } hdr[MAXR];
};
struct Op {
  unsigned long long seq;
  size_t send_cap = 0, recv_cap = 0;  // sizes of the pinned staging buffers
  size_t bytes;      // per rank
  size_t count;
  int dtype;
  char *hsend, *hrecv;  // pinned staging of this op
  hipEvent_t sent, done;
};
struct Comm {
  int rank, nranks, device;
  char name[64];
  size_t bytes;
  Shared *sh;
  char *slots;
  // asynchronous part
  unsigned long long seq = 0;
  unsigned long long *flag_h = nullptr, *flag_d = nullptr;  // mapped host word: last finished op
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Op *> queue;     // enqueued, not yet exchanged
  std::deque<Op *> retired;   // exchanged; staging is free once `done` has completed
  bool stop = false;
  bool sync_mode = false;     // LBFGSB_FAKE_RCCL_SYNC
  std::atomic<int> error{0};  // sticky: a count / datatype mismatch was seen
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

// one wave that waits until the worker has finished op `seq` (the word lives in mapped host memory);
// bounded: every wave leaves after 2^23 polls (tens of seconds) whatever happens, so the grid always drains
__global__ void wait_exchange_kernel(const unsigned long long *flag, unsigned long long seq) {
  for (unsigned it = 0; it < (1u << 23); ++it) {
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= seq) return;
    __builtin_amdgcn_s_sleep(64);
  }
}

// the exchange of one collective: publish count / datatype, deposit, meet, check, gather
void exchange(Comm *c, Op *op) {
  c->sh->hdr[c->rank].count = op->count;
  c->sh->hdr[c->rank].dtype = op->dtype;
  std::memcpy(c->slots + (size_t)c->rank * SLOT, op->hsend, op->bytes);
  barrier(c);
  bool bad = false;
  for (int r = 0; r < c->nranks; ++r)
    if (c->sh->hdr[r].count != op->count || c->sh->hdr[r].dtype != op->dtype) {
      if (!bad)
        std::fprintf(stderr,
                     "fake_rccl: rank %d, collective %llu: rank %d passed count %llu / datatype %d, this rank "
                     "%zu / %d -- real RCCL would hang or corrupt here\n",
                     c->rank, op->seq, r, c->sh->hdr[r].count, c->sh->hdr[r].dtype, op->count, op->dtype);
      bad = true;
    }
  if (bad) {
    c->error.store((int)ncclInvalidArgument);
    double *o = reinterpret_cast<double *>(op->hrecv);
    for (size_t k = 0; k < (size_t)c->nranks * op->bytes / sizeof(double); ++k) o[k] = std::nan("");
  } else {
    for (int r = 0; r < c->nranks; ++r)
      std::memcpy(op->hrecv + (size_t)r * op->bytes, c->slots + (size_t)r * SLOT, op->bytes);
  }
  barrier(c);  // everyone has read the slots before they are reused
}

void run_worker(Comm *c) {
  (void)hipSetDevice(c->device);
  for (;;) {
    Op *op = nullptr;
    {
      std::unique_lock<std::mutex> lk(c->mu);
      c->cv.wait(lk, [&] { return c->stop || !c->queue.empty(); });
      if (c->queue.empty()) return;  // stop, nothing left
      op = c->queue.front();
      c->queue.pop_front();
    }
    (void)hipEventSynchronize(op->sent);  // this rank's send buffer has reached the staging area
    exchange(c, op);
    __atomic_store_n(c->flag_h, op->seq, __ATOMIC_RELEASE);  // the stream's wait kernel goes on
    {
      std::lock_guard<std::mutex> lk(c->mu);
      c->retired.push_back(op);
    }
  }
}

// staging buffers and events are reused: a retired op whose H2D copy has run serves the next collective
// that fits (pinned allocations cost ~0.1-1 ms each; the solver issues thousands of collectives per test)
void free_op(Op *op) {
  (void)hipHostFree(op->hsend);
  (void)hipHostFree(op->hrecv);
  (void)hipEventDestroy(op->sent);
  (void)hipEventDestroy(op->done);
  delete op;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0, sizeof *id);
  std::snprintf(id->internal, sizeof id->internal, "/lbfgsb_fake_rccl_%d_%ld", (int)getpid(), random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm *c = new Comm;
  c->rank = rank, c->nranks = nranks;
  if (hipGetDevice(&c->device) != hipSuccess) return ncclUnhandledCudaError;
  std::snprintf(c->name, sizeof c->name, "%s", id.internal);
  c->bytes = 8192 + (size_t)nranks * SLOT;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError;
  void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  static_assert(sizeof(Shared) <= 8192, "header page");
  c->sh = static_cast<Shared *>(p);  // (a fresh segment is zero-filled: counters start at 0)
  c->slots = static_cast<char *>(p) + 8192;
  if (hipHostMalloc(&c->flag_h, 64) != hipSuccess) return ncclUnhandledCudaError;
  *c->flag_h = 0;
  if (hipHostGetDevicePointer((void **)&c->flag_d, c->flag_h, 0) != hipSuccess) return ncclUnhandledCudaError;
  const char *sm = std::getenv("LBFGSB_FAKE_RCCL_SYNC");
  c->sync_mode = sm && sm[0] == '1';
  if (!c->sync_mode) c->worker = std::thread(run_worker, c);
  *out = reinterpret_cast<ncclComm_t>(c);
  barrier(c);
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
  *count = reinterpret_cast<const Comm *>(comm)->nranks;
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
  *rank = reinterpret_cast<const Comm *>(comm)->rank;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  (void)hipDeviceSynchronize();  // (every enqueued collective has run)
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->stop = true;
  }
  c->cv.notify_all();
  if (c->worker.joinable()) c->worker.join();
  for (Op *op : c->retired) free_op(op);
  for (Op *op : c->queue) free_op(op);
  barrier(c);
  const int err = c->error.load();
  munmap(c->sh, c->bytes);
  if (c->rank == 0) shm_unlink(c->name);
  (void)hipHostFree(c->flag_h);
  delete c;
  return err ? (ncclResult_t)err : ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (c->error.load()) return (ncclResult_t)c->error.load();  // a mismatch earlier on this communicator
  size_t esz = dt == ncclDouble || dt == ncclInt64 || dt == ncclUint64 ? 8 : dt == ncclFloat || dt == ncclInt32 ? 4 : 0;
  if (esz != 8) return ncclInvalidArgument;  // (the poison value and the staging assume 8-byte elements)
  if (count * esz > SLOT) return ncclInvalidArgument;
  const size_t nbytes = count * esz;
  const size_t sb = nbytes ? nbytes : 8, rb = (size_t)c->nranks * sb;
  Op *op = nullptr;
  {  // the staging of a finished collective whose H2D copy has run is free again: take the first that fits.
     // Nothing is ever FREED here (hipHostFree synchronises the device: with thread ranks it would wait for
     // the other ranks' stream-side waits); the pool goes at ncclCommDestroy
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto it = c->retired.begin(); it != c->retired.end(); ++it)
      if ((*it)->send_cap >= sb && (*it)->recv_cap >= rb && hipEventQuery((*it)->done) == hipSuccess) {
        op = *it;
        c->retired.erase(it);
        break;
      }
  }
  if (!op) {
    op = new Op;
    op->send_cap = sb < 4096 ? 4096 : sb, op->recv_cap = rb < 65536 ? 65536 : rb;
    if (hipHostMalloc(&op->hsend, op->send_cap) != hipSuccess || hipHostMalloc(&op->hrecv, op->recv_cap) != hipSuccess)
      return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&op->sent, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&op->done, hipEventDisableTiming) != hipSuccess)
      return ncclUnhandledCudaError;
  }
  op->seq = ++c->seq, op->count = count, op->dtype = (int)dt, op->bytes = nbytes;
  if (c->sync_mode) {  // thread ranks: everything inside the call (see the header)
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (op->bytes && hipMemcpy(op->hsend, send, op->bytes, hipMemcpyDeviceToHost) != hipSuccess)
      return ncclUnhandledCudaError;
    exchange(c, op);
    if (op->bytes && hipMemcpy(recv, op->hrecv, (size_t)c->nranks * op->bytes, hipMemcpyHostToDevice) != hipSuccess)
      return ncclUnhandledCudaError;
    (void)hipEventRecord(op->done, stream);
    {
      std::lock_guard<std::mutex> lk(c->mu);
      c->retired.push_back(op);
    }
    return c->error.load() ? (ncclResult_t)c->error.load() : ncclSuccess;
  }
  // everything below is ENQUEUED: the call returns before a byte has moved
  if (op->bytes && hipMemcpyAsync(op->hsend, send, op->bytes, hipMemcpyDeviceToHost, stream) != hipSuccess)
    return ncclUnhandledCudaError;
  if (hipEventRecord(op->sent, stream) != hipSuccess) return ncclUnhandledCudaError;
  hipLaunchKernelGGL(wait_exchange_kernel, dim3(1), dim3(1), 0, stream, c->flag_d, op->seq);
  if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
  if (op->bytes && hipMemcpyAsync(recv, op->hrecv, (size_t)c->nranks * op->bytes, hipMemcpyHostToDevice, stream) !=
                       hipSuccess)
    return ncclUnhandledCudaError;
  if (hipEventRecord(op->done, stream) != hipSuccess) return ncclUnhandledCudaError;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->queue.push_back(op);
  }
  c->cv.notify_one();
  return ncclSuccess;
}

}  // extern "C"
