// hbm_write_peak.hip — what a kernel that only WRITES (or only reads) reaches on this GPU: the bound of the keep-partials evaluation,
// whose per-tree kernel stores 7.2 GB of partials per launch at the benchmark's size (bench.py: fallbacks, PAML_AMD_KEEP_PARTIALS).
// 16-byte accesses, 1 KB per wave instruction, every workgroup a contiguous 64 KB piece per step — the pattern of jit_store.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_write_peak tools/hbm_write_peak.hip && tools/hbm_write_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE>      // 0: plain stores, 1: nontemporal stores, 2: loads
__global__ __launch_bounds__(512) void k(d2 *p, size_t n16, double *sink)
{
   const size_t stride = (size_t)gridDim.x * 512;
   d2 acc = {0, 0};
   for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += stride) {
      if (MODE == 0) p[i] = (d2){1.0, 2.0};
      else if (MODE == 1) __builtin_nontemporal_store((d2){1.0, 2.0}, p + i);
      else acc += p[i];
   }
   if (MODE == 2 && acc.x == 123.456) *sink = acc.y;
}
int main()
{
   const size_t bytes = (size_t)7168 << 20, n16 = bytes / 16;
   d2 *p; double *sink;
   if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { puts("hipMalloc failed"); return 1; }
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const char *names[3] = {"stores", "nontemporal stores", "loads"};
   for (int mode = 0; mode < 3; mode++)
      for (int grid : {256, 1024, 4096}) {
         float best = 1e9;
         for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, p, n16, sink);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, p, n16, sink);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, p, n16, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
         }
         printf("%-20s grid %5d: %.3f ms for %.2f GB = %.2f TB/s\n", names[mode], grid, best, bytes / 1e9, bytes / 1e9 / best);
      }
   return 0;
}
