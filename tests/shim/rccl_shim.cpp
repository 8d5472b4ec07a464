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
//
// DEVICE MODE (PAML_AMD_SHIM_DEVICE_US=T, T >= 0): ncclAllReduce behaves like the real collective towards the GPU and the host — it
// returns at once and launches a KERNEL on the caller's stream (PAML_AMD_SHIM_WGS workgroups, default 1, of 512 threads with 64 KB of
// LDS each: it needs a CU of its own the way RCCL's LL kernel does, it cannot slip in beside a workgroup that fills a CU).  The kernel
// copies the rank's values into the shared segment (host memory mapped into every rank's GPU address space), raises the rank's
// arrival flag, SPINS until every rank's flag is up and at least T microseconds have passed since it started (the xGMI round trips of
// an 8-rank all-reduce of a few KB: 10 .. 40 us), and adds the ranks' arrays in rank order.  With one rank it is the T-microsecond CU
// occupant alone.  Two sets of slots alternate by call parity: a rank enters call s + 2 only after every rank has arrived at s + 1,
// i.e. finished reading the slots of call s (stream order).  A rank that never arrives ends the spin after 2 s with the segment's error
// flag set (ncclCommDestroy then reports it) instead of hanging the GPU.  What profiles/r04_comm_emulated.txt was measured with.
// Build: hipcc -O2 -shared -fPIC --offload-arch=gfx950 -o librccl_shim.so rccl_shim.cpp -lrt
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
   long dev_arrive[MAX_RANKS];             // device mode: written and polled by the kernels (system-scope atomics)
   int dev_error;                          // device mode: a spin timed out
   double slot[MAX_RANKS][MAX_COUNT];
   double dev_slot[2][MAX_RANKS][MAX_COUNT];
};

struct Comm {
   int rank, world;
   long seq;
   Shared *sh;
   Shared *dsh;                            // device mode: the segment as the GPU sees it (hipHostRegister)
   int dev_us, dev_wgs;                    // device mode: -1 = host mode
   char name[64];
};

// Device mode: see the header.  Block 0 does the exchange; further blocks only occupy a CU for the same time.
__global__ __launch_bounds__(512) void shim_allreduce_kernel(Shared *sh, int rank, int world, long seq, const double *send, double *recv,
                                                             int count, long ticks, long timeout_ticks)
{
   extern __shared__ double lds_pad[];      // 64 KB: never touched, it is there to claim the CU's LDS like a collective kernel's FIFOs
   const long t0 = wall_clock64();          // 100 MHz
   const int par = (int)(seq & 1);
   if (blockIdx.x == 0) {
      for (int i = threadIdx.x; i < count; i += blockDim.x)
         __hip_atomic_store(&sh->dev_slot[par][rank][i], send[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) {
         __hip_atomic_store(&sh->dev_arrive[rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
         for (int r = 0; r < world; r++)
            while (__hip_atomic_load(&sh->dev_arrive[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
               __builtin_amdgcn_s_sleep(8);
               if (wall_clock64() - t0 > timeout_ticks) { __hip_atomic_store(&sh->dev_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
      }
   }
   if (threadIdx.x == 0)
      while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
   __syncthreads();
   if (blockIdx.x == 0) {
      __threadfence_system();
      for (int i = threadIdx.x; i < count; i += blockDim.x) {
         double s = __hip_atomic_load(&sh->dev_slot[par][0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
         for (int r = 1; r < world; r++) s += __hip_atomic_load(&sh->dev_slot[par][r][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
         recv[i] = s;
      }
   }
   if (threadIdx.x == 9999) lds_pad[0] = 0;
}

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
   c->dsh = nullptr; c->dev_us = -1; c->dev_wgs = 1;
   if (const char *v = getenv("PAML_AMD_SHIM_DEVICE_US")) {
      c->dev_us = atoi(v) < 0 ? 0 : atoi(v);
      if (const char *w = getenv("PAML_AMD_SHIM_WGS")) c->dev_wgs = atoi(w) < 1 ? 1 : atoi(w);
      void *dp = nullptr;
      if (hipHostRegister(p, sizeof(Shared), hipHostRegisterMapped) != hipSuccess || hipHostGetDevicePointer(&dp, p, 0) != hipSuccess ||
          hipFuncSetAttribute((const void *)shim_allreduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) {
         munmap(p, sizeof(Shared)); delete c; return ncclUnhandledCudaError;
      }
      c->dsh = (Shared *)dp;
   }
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
   ncclResult_t res = ncclSuccess;
   if (c->dsh) {
      (void)hipDeviceSynchronize();
      if (c->sh->dev_error) res = ncclSystemError;      // a device-mode spin gave up waiting for a rank
      (void)hipHostUnregister(c->sh);
   }
   munmap(c->sh, sizeof(Shared));
   delete c;
   return res;
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
   Comm *c = (Comm *)comm;
   if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
   if (datatype != ncclDouble || op != ncclSum || count > MAX_COUNT) return ncclInvalidUsage;
   Shared *sh = c->sh;
   const long seq = ++c->seq;
   if (c->dsh) {      // device mode: a kernel on the caller's stream, the call returns at once
      hipLaunchKernelGGL(shim_allreduce_kernel, dim3(c->dev_wgs), dim3(512), 64 * 1024, stream, c->dsh, c->rank, c->world, seq, (const double *)sendbuff,
                         (double *)recvbuff, (int)count, (long)c->dev_us * 100, 200000000L);
      return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
   }
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
