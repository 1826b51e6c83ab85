// rccl_stub.cpp -- TEST INFRASTRUCTURE, not part of the product.
//
// A stand-in for the six RCCL entry points csrc/comm.hip binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclCommAbort, ncclAllGather, ncclBroadcast, ncclGetErrorString) that lets SEVERAL ranks share ONE GPU: RCCL itself
// refuses two ranks per device, so on the 1-GPU test box the W > 1 branches of az_comm_gather_push (padded segments,
// per-rank offsets, re-ordering by global game id) and az_comm_broadcast_params could never execute.  The library is
// selected with AZHIP_RCCL_LIB (csrc/comm.hip, rc::load) by tests/test_comm_stub_gpu.py only.
//
// Transport: one file in /dev/shm (or /tmp) per communicator, mapped by every rank: a header with a generation barrier
// and one slot per rank.  A collective = wait for the stream, device -> own slot, barrier, slots -> device, barrier.
// Blocking and slow on purpose; every wait is bounded (a missing rank gives ncclSystemError, not a hang).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclDataType_t;
}

namespace {
struct Header {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> attached;
  uint32_t nranks;
  uint64_t slot_bytes;
};
struct Comm {
  Header* h;
  char* slots;
  size_t map_bytes, slot_bytes;
  int rank, nranks;
  char path[192];
  double timeout_s;
};
size_t dtype_size(int t) {
  switch (t) {
    case 0: case 1: case 10: case 11: return 1;
    case 6: case 9: return 2;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
  }
  return 0;
}
bool barrier(Comm* c) {
  const uint32_t gen = c->h->generation.load(std::memory_order_acquire);
  if (c->h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
    c->h->arrived.store(0, std::memory_order_relaxed);
    c->h->generation.store(gen + 1, std::memory_order_release);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (c->h->generation.load(std::memory_order_acquire) == gen) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return true;
}
}  // namespace

extern "C" const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "stub: HIP call failed";
    case ncclSystemError: return "stub: a rank did not arrive (timeout) or the shared file could not be mapped";
    case ncclInvalidArgument: return "stub: invalid argument";
    default: return "stub: internal error";
  }
}

extern "C" ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  unsigned long long r = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(id->internal, sizeof id->internal, "azstub-%d-%llx", (int)getpid(), r);
  return ncclSuccess;
}

extern "C" ncclResult_t ncclCommInitRank(void** out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || strncmp(id.internal, "azstub-", 7) != 0) return ncclInvalidArgument;
  Comm* c = new Comm();
  c->rank = rank; c->nranks = nranks;
  const char* mb = getenv("AZSTUB_SLOT_MB");
  c->slot_bytes = (size_t)(mb ? atoi(mb) : 64) << 20;
  const char* to = getenv("AZSTUB_TIMEOUT_S");
  c->timeout_s = to ? atof(to) : 120.0;
  struct stat sb;
  id.internal[100] = 0;
  snprintf(c->path, sizeof c->path, "%s/%s", stat("/dev/shm", &sb) == 0 ? "/dev/shm" : "/tmp", id.internal);
  c->map_bytes = 4096 + c->slot_bytes * (size_t)nranks;
  int fd = open(c->path, O_RDWR | O_CREAT, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);   // a fresh file reads as zeros: counters start at 0
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->h = (Header*)p;
  c->slots = (char*)p + 4096;
  c->h->attached.fetch_add(1);
  const bool ok = barrier(c);
  if (rank == 0) unlink(c->path);                      // everybody has it mapped (or gave up): nothing is left behind
  if (!ok) { munmap(p, c->map_bytes); delete c; return ncclSystemError; }
  *out = c;
  return ncclSuccess;
}

extern "C" ncclResult_t ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclSuccess;
  munmap((void*)c->h, c->map_bytes);
  delete c;
  return ncclSuccess;
}
extern "C" ncclResult_t ncclCommAbort(void* comm) { return ncclCommDestroy(comm); }

extern "C" ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t bytes = count * dtype_size(dt);
  if (!c || !dtype_size(dt)) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  char* own = c->slots + c->slot_bytes * (size_t)c->rank;
  for (size_t off = 0; off < bytes || off == 0; off += c->slot_bytes) {          // messages larger than a slot go in pieces
    const size_t n = bytes - off < c->slot_bytes ? bytes - off : c->slot_bytes;
    if (n && hipMemcpy(own, (const char*)send + off, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->nranks && n; ++r)
      if (hipMemcpy((char*)recv + (size_t)r * bytes + off, c->slots + c->slot_bytes * (size_t)r, n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    if (!bytes) break;
  }
  return ncclSuccess;
}

extern "C" ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  const size_t bytes = count * dtype_size(dt);
  if (!c || !dtype_size(dt) || root < 0 || root >= c->nranks) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  for (size_t off = 0; off < bytes || off == 0; off += c->slot_bytes) {
    const size_t n = bytes - off < c->slot_bytes ? bytes - off : c->slot_bytes;
    if (c->rank == root && n && hipMemcpy(c->slots, (const char*)send + off, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    if (n && (c->rank != root || send != recv) && hipMemcpy((char*)recv + off, c->slots, n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    if (!bytes) break;
  }
  return ncclSuccess;
}
