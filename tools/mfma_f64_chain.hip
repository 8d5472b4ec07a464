// Latency of a DEPENDENT chain of v_mfma_f64_16x16x4_f64 (each instruction's accumulator is the previous one's result), one wave per
// SIMD, against 2 and 4 interleaved independent chains: what a product costs in prune_mfma64_coop (one accumulator per wave and
// branch).  hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_chain tools/mfma_f64_chain.hip && tools/mfma_f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void chain(double *out, long long *cyc, int n)
{
   v4d acc[CH];
   for (int c = 0; c < CH; c++) acc[c] = (v4d){0, 0, 0, 0};
   double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
   long long t0 = __builtin_amdgcn_s_memtime();
   for (int i = 0; i < n; i++) {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
         for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
   }
   long long t1 = __builtin_amdgcn_s_memtime();
   double s = 0;
   for (int c = 0; c < CH; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
   out[blockIdx.x * blockDim.x + threadIdx.x] = s;
   if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
static void run(const char *what)
{
   double *out; long long *cyc, h;
   hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
   const int n = 1000;
   chain<CH><<<1, 64>>>(out, cyc, n);
   chain<CH><<<1, 64>>>(out, cyc, n);
   hipDeviceSynchronize();
   hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
   printf("%-28s %6.1f s_memtime ticks per MFMA (%d MFMAs)\n", what, (double)h / (16.0 * n * CH), 16 * n * CH);
   hipFree(out); hipFree(cyc);
}
int main()
{
   run<1>("1 dependent chain");
   run<2>("2 interleaved chains");
   run<4>("4 interleaved chains");
   return 0;
}
