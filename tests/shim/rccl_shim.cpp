// rccl_shim.cpp — TEST INFRASTRUCTURE ONLY: the five RCCL entry points libpaml_amd.so binds (engine_state.h: Rccl), implemented over
// POSIX shared memory, so that world > 1 can be exercised with every rank on ONE GPU (the GPU box of the test tier has one; real
// RCCL refuses two ranks on the same device).  Selected with PAML_AMD_RCCL_LIB=<this library>; nothing in the product links it.
//
//   ncclGetUniqueId     a fresh shared-memory segment, its name inside the 128-byte id
//   ncclCommInitRank    maps the segment, arrival barrier
//   ncclAllReduce       ncclDouble + ncclSum only: device -> host slot of this rank, barrier, sum over the ranks IN RANK ORDER, host ->
//                       device, barrier.  Blocks the calling host thread until every rank has arrived (a real collective only
//                       blocks the stream); the engine's results do not depend on that.
//   ncclCommDestroy     unmaps; rank 0 unlinks
// Build: hipcc -O2 -shared -fPIC -o librccl_shim.so rccl_shim.cpp -lrt
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int MAX_RANKS = 8;
constexpr size_t MAX_COUNT = 1 << 16;      // doubles per rank and call (the engine sends <= ~1024 x batch partial sums)

struct Shared {
   std::atomic<int> ready;                 // rank 0 has initialised the header
   std::atomic<int> joined;
   std::atomic<long> arrive[MAX_RANKS], depart[MAX_RANKS];
   double slot[MAX_RANKS][MAX_COUNT];
};

struct Comm {
   int rank, world;
   long seq;
   Shared *sh;
   char name[64];
};

const char *MAGIC = "paml_amd_rccl_shim:";

bool wait_all(std::atomic<long> *a, int world, long seq)
{
   const time_t t0 = time(nullptr);
   for (int r = 0; r < world; r++)
      while (a[r].load(std::memory_order_acquire) < seq) {
         sched_yield();
         if (time(nullptr) - t0 > 120) return false;      // a rank died: fail instead of hanging the test box
      }
   return true;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
   if (!id) return ncclInvalidArgument;
   memset(id, 0, sizeof(*id));
   char name[64];
   snprintf(name, sizeof(name), "/paml_amd_shim_%d_%ld", (int)getpid(), (long)time(nullptr) ^ (long)rand());
   const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
   if (fd < 0) return ncclSystemError;
   if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); shm_unlink(name); return ncclSystemError; }
   void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
   close(fd);
   if (p == MAP_FAILED) { shm_unlink(name); return ncclSystemError; }
   Shared *sh = new (p) Shared();      // (ftruncate zero-filled it; the atomics start at 0)
   sh->ready.store(1, std::memory_order_release);
   munmap(p, sizeof(Shared));
   snprintf(id->internal, sizeof(id->internal), "%s%s", MAGIC, name);
   return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
   if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
   if (strncmp(id.internal, MAGIC, strlen(MAGIC)) != 0) return ncclInvalidArgument;
   Comm *c = new Comm();
   c->rank = rank; c->world = nranks; c->seq = 0;
   snprintf(c->name, sizeof(c->name), "%s", id.internal + strlen(MAGIC));
   const int fd = shm_open(c->name, O_RDWR, 0600);
   if (fd < 0) { delete c; return ncclSystemError; }
   void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
   close(fd);
   if (p == MAP_FAILED) { delete c; return ncclSystemError; }
   c->sh = (Shared *)p;
   c->sh->joined.fetch_add(1, std::memory_order_acq_rel);
   const time_t t0 = time(nullptr);
   while (c->sh->joined.load(std::memory_order_acquire) < nranks) {      // like the real call: returns when every rank has joined
      sched_yield();
      if (time(nullptr) - t0 > 120) { munmap(p, sizeof(Shared)); delete c; return ncclSystemError; }
   }
   *comm = (ncclComm_t)c;
   return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
   Comm *c = (Comm *)comm;
   if (!c) return ncclInvalidArgument;
   if (c->rank == 0) shm_unlink(c->name);
   munmap(c->sh, sizeof(Shared));
   delete c;
   return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
   Comm *c = (Comm *)comm;
   if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
   if (datatype != ncclDouble || op != ncclSum || count > MAX_COUNT) return ncclInvalidUsage;
   Shared *sh = c->sh;
   const long seq = ++c->seq;
   if (hipMemcpyAsync(sh->slot[c->rank], sendbuff, count * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
   if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
   sh->arrive[c->rank].store(seq, std::memory_order_release);
   if (!wait_all(sh->arrive, c->world, seq)) return ncclSystemError;
   std::vector<double> sum(sh->slot[0], sh->slot[0] + count);
   for (int r = 1; r < c->world; r++)
      for (size_t i = 0; i < count; i++) sum[i] += sh->slot[r][i];
   if (hipMemcpyAsync(recvbuff, sum.data(), count * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
   if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
   sh->depart[c->rank].store(seq, std::memory_order_release);      // the slots may be overwritten once every rank has read them
   if (!wait_all(sh->depart, c->world, seq)) return ncclSystemError;
   return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
   switch (r) {
   case ncclSuccess: return "no error";
   case ncclInvalidArgument: return "shim: invalid argument";
   case ncclInvalidUsage: return "shim: only ncclDouble / ncclSum, at most 2^16 elements";
   case ncclSystemError: return "shim: shared memory or a rank that never arrived";
   default: return "shim: HIP error";
   }
}

}  // extern "C"
