// tools/mfma4_layout.hip — finds the operand layout of v_mfma_f64_4x4x4_4b_f64 on the box it runs on (gfx950): for every pair
// (lane la of A holding 1, lane lb of B holding 1, all else 0) it records which D lanes become 1.  From the table: which lanes
// form a block, and which of the 16 lanes of a block hold row i / inner index k (A), inner k / column j (B), row i / column j (D).
// build: hipcc --offload-arch=gfx950 -O2 -o mfma4_layout tools/mfma4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned long long *out)
{
   const int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
   const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
   const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
   const unsigned long long m = __ballot(d != 0.0);
   if (lane == 0) out[la * 64 + lb] = m;
}
int main()
{
   unsigned long long *d;
   (void)hipMalloc(&d, 64 * 64 * 8);
   hipLaunchKernelGGL(k, dim3(64, 64), dim3(64), 0, 0, d);
   std::vector<unsigned long long> h(64 * 64);
   (void)hipMemcpy(h.data(), d, 64 * 64 * 8, hipMemcpyDeviceToHost);
   // for each A lane: the set of B lanes it pairs with (non-zero product) and the D lanes hit
   for (int la = 0; la < 64; la++) {
      printf("A lane %2d:", la);
      for (int lb = 0; lb < 64; lb++)
         if (h[la * 64 + lb]) {
            printf("  B%-2d->D", lb);
            for (int l = 0; l < 64; l++)
               if (h[la * 64 + lb] >> l & 1) printf("%d,", l);
         }
      printf("\n");
   }
   return 0;
}
