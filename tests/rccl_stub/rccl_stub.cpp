// TEST-ONLY stand-in for librccl: the six entry points csrc/comm.hip binds at run time, implemented over POSIX shared memory
// between processes that may share ONE GPU (RCCL itself refuses two ranks on one device, and the gpurun boxes have one).
// Selected by the tests through SEGVLAD_RCCL_LIB (the library's own override of the RCCL it dlopens); never loaded otherwise.
//
//   ncclAllGather   stream-ordered for the caller: synchronises the stream, copies the send buffer into the rank's slot of the
//                   shared segment, meets the other ranks at a barrier, copies all slots into the receive buffer, meets them
//                   again (nobody overwrites a slot somebody still reads).
//   ncclCommAbort   marks the segment aborted: every rank waiting at (or arriving at) a barrier returns an error instead of
//                   waiting -- the behaviour comm.hip relies on when one rank fails before a collective.
//   A barrier also gives up after SVSTUB_TIMEOUT_S seconds (default 20): a test that would hang fails instead.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

namespace {
constexpr size_t SLOT_BYTES = 48u << 20;   // per rank
constexpr int MAX_RANKS = 8;
struct Header {
  std::atomic<uint32_t> arrived[2];   // two alternating barrier counters
  std::atomic<uint32_t> generation;   // completed barriers
  std::atomic<uint32_t> aborted;
  std::atomic<uint32_t> attached;
};
struct Comm {
  int rank, world;
  char name[64];
  unsigned char* base;   // Header, then world slots
  size_t bytes;
  uint32_t barriers;     // barriers this rank has passed
};
double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}
double timeout_s() {
  const char* e = getenv("SVSTUB_TIMEOUT_S");
  return e ? atof(e) : 20.0;
}
// 0 = passed, 2 = aborted / timed out
int barrier(Comm* c) {
  Header* h = reinterpret_cast<Header*>(c->base);
  const uint32_t gen = c->barriers;
  std::atomic<uint32_t>& cnt = h->arrived[gen & 1];
  if (h->aborted.load()) return 2;
  if (cnt.fetch_add(1) + 1 == (uint32_t)c->world) {
    cnt.store(0);
    h->generation.store(gen + 1);
  } else {
    const double t0 = now_s();
    while (h->generation.load() == gen) {
      if (h->aborted.load()) return 2;
      if (now_s() - t0 > timeout_s()) {
        h->aborted.store(1);
        return 2;
      }
      usleep(50);
    }
  }
  c->barriers = gen + 1;
  return 0;
}
size_t dtype_bytes(int t) {
  switch (t) {
    case 0: case 1: return 1;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: return 2;
    default: return 0;
  }
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/svstub_%d_%llx", (int)getpid(), (unsigned long long)(now_s() * 1e6));
  return 0;
}

int ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return 4;   // ncclInvalidArgument
  Comm* c = new Comm();
  c->rank = rank;
  c->world = nranks;
  c->barriers = 0;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->bytes = 4096 + (size_t)nranks * SLOT_BYTES;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
    delete c;
    return 2;   // ncclSystemError
  }
  c->base = static_cast<unsigned char*>(mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->base == MAP_FAILED) {
    delete c;
    return 2;
  }
  reinterpret_cast<Header*>(c->base)->attached.fetch_add(1);   // (a fresh segment is all zero)
  if (barrier(c) != 0) {   // the init is a collective, like RCCL's
    munmap(c->base, c->bytes);
    delete c;
    return 2;
  }
  *out = c;
  return 0;
}

static int detach(Comm* c) {
  Header* h = reinterpret_cast<Header*>(c->base);
  const bool last = h->attached.fetch_sub(1) == 1;
  munmap(c->base, c->bytes);
  if (last) shm_unlink(c->name);
  delete c;
  return 0;
}

int ncclCommDestroy(ncclComm_t comm) { return comm ? detach(static_cast<Comm*>(comm)) : 0; }

int ncclCommAbort(ncclComm_t comm) {
  if (!comm) return 0;
  Comm* c = static_cast<Comm*>(comm);
  reinterpret_cast<Header*>(c->base)->aborted.store(1);
  return detach(c);
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, ncclComm_t comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t bytes = count * dtype_bytes(dtype);
  if (!c || bytes == 0 || bytes > SLOT_BYTES) return 4;
  unsigned char* slots = c->base + 4096;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;   // ncclUnhandledCudaError
  if (hipMemcpy(slots + (size_t)c->rank * SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (barrier(c) != 0) return 2;
  for (int r = 0; r < c->world; ++r)
    if (hipMemcpy(static_cast<unsigned char*>(recv) + (size_t)r * bytes, slots + (size_t)r * SLOT_BYTES, bytes, hipMemcpyHostToDevice) !=
        hipSuccess)
      return 1;
  if (barrier(c) != 0) return 2;
  return 0;
}

const char* ncclGetErrorString(int r) {
  switch (r) {
    case 0: return "no error";
    case 1: return "unhandled HIP error (rccl test stub)";
    case 2: return "system error: peer aborted or timed out (rccl test stub)";
    case 4: return "invalid argument (rccl test stub)";
    default: return "error (rccl test stub)";
  }
}
}
