// hbm_mix_peak.hip — what a kernel with the traffic of the coefficient-forming kernel (branch_eig_kernel, kernels_branch.h: two 0.5 GB
// arrays of partials read, one 0.5 GB array of coefficients written, 512 B per pattern each) reaches on this GPU when it does nothing else:
// C = A * B elementwise, 16-byte non-temporal accesses, a wave takes 8 KB pieces (one 16-pattern group of 64 states) of A, B and C as the
// forming kernel's waves do, DEPTH pieces requested ahead, one or two workgroups of 512 threads per CU.  The practical ceiling of that
// kernel's memory side (its matrix-pipe side is as long: 14 884 flop per pattern = 0.189 ms against 0.192 ms at 8 TB/s).
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_mix_peak tools/hbm_mix_peak.hip && tools/hbm_mix_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int DEPTH, bool NT>
__global__ __launch_bounds__(512) void mix(const d2 *A, const d2 *B, d2 *C, long n_groups)
{
   const int lane = threadIdx.x & 63;
   const long wave = (long)blockIdx.x * 8 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 8;
   d2 a[DEPTH][8], b[DEPTH][8];
   auto ld = [&](long g, d2 (&x)[8], const d2 *P) {
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = NT ? __builtin_nontemporal_load(P + g * 512 + i * 64 + lane) : P[g * 512 + i * 64 + lane];
   };
   long g = wave;
#pragma unroll
   for (int d = 0; d < DEPTH; d++) {
      const long gd = g + d * n_waves < n_groups ? g + d * n_waves : g;
      ld(gd, a[d], A); ld(gd, b[d], B);
   }
   for (; g < n_groups; g += DEPTH * n_waves) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
         const long gc = g + d * n_waves;
         if (gc >= n_groups) break;
         d2 c[8];
#pragma unroll
         for (int i = 0; i < 8; i++) c[i] = a[d][i] * b[d][i];
         const long gn = gc + DEPTH * n_waves < n_groups ? gc + DEPTH * n_waves : gc;
         ld(gn, a[d], A); ld(gn, b[d], B);
#pragma unroll
         for (int i = 0; i < 8; i++) {
            if (NT) __builtin_nontemporal_store(c[i], C + gc * 512 + i * 64 + lane);
            else C[gc * 512 + i * 64 + lane] = c[i];
         }
      }
   }
}
int main()
{
   const long n_patt = 1000000, n_groups = (n_patt + 15) / 16;
   const size_t bytes = (size_t)n_groups * 8192;
   d2 *A, *B, *C;
   if (hipMalloc(&A, bytes) != hipSuccess || hipMalloc(&B, bytes) != hipSuccess || hipMalloc(&C, bytes) != hipSuccess) { puts("hipMalloc failed"); return 1; }
   hipMemset(A, 0, bytes); hipMemset(B, 0, bytes);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   auto run = [&](const char *name, auto kern, int grid) {
      float best = 1e9;
      for (int rep = 0; rep < 6; rep++) {
         hipEventRecord(e0);
         hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, A, B, C, n_groups);
         hipEventRecord(e1); hipEventSynchronize(e1);
         float ms; hipEventElapsedTime(&ms, e0, e1);
         if (rep && ms < best) best = ms;
      }
      printf("%-34s grid %5d: %.4f ms for %.3f GB = %.2f TB/s (%.3f of 8 TB/s)\n", name, grid, best, 3 * bytes / 1e9, 3 * bytes / 1e9 / best, 3 * bytes / 1e9 / best / 8.0);
   };
   for (int grid : {256, 512, 1024, 2048}) {
      run("depth 1, non-temporal", mix<1, true>, grid);
      run("depth 2, non-temporal", mix<2, true>, grid);
      run("depth 4, non-temporal", mix<4, true>, grid);
      run("depth 2, plain", mix<2, false>, grid);
   }
   return 0;
}
