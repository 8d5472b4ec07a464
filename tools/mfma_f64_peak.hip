// tools/mfma_f64_peak.hip — measures the FP64 ceilings of the box this runs on (MI355X, gfx950):
//   (a) v_mfma_f64_16x16x4_f64 issue rate (4 independent accumulators per wave),
//   (b) v_fma_f64 VALU rate, (c) both pipes fed at once from different waves of the same SIMD.
// The result anchors the `peak` of bench.py's roofline.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak tools/mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: mfma only, 1: valu only, 2: even waves mfma / odd waves valu, 3: v_mfma_f64_4x4x4_4b_f64 only
__global__ __launch_bounds__(512) void k(double *out, int iters, double seed)
{
   const int wave = threadIdx.x >> 6;
   // shader clock over the kernel's run: s_memtime (core clock) against s_memrealtime (100 MHz), workgroup 0
   const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
   struct Stamp { double *o; unsigned long long c0, r0; __device__ ~Stamp() { if (blockIdx.x == 0 && threadIdx.x == 0) { o[1] = (double)(__builtin_amdgcn_s_memtime() - c0); o[2] = (double)(__builtin_amdgcn_s_memrealtime() - r0); } } } stamp{out, c0, r0};
   if (MODE == 3) {      // four 4x4x4 blocks per instruction: 4 x 64 MACs = 512 FLOP
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
      double x = seed + threadIdx.x * 1e-9, y = 1.0 - seed;
      for (int i = 0; i < iters; i++) {
#pragma unroll
         for (int u = 0; u < 4; u++) {
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a4, 0, 0, 0);
            a5 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a5, 0, 0, 0);
            a6 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a6, 0, 0, 0);
            a7 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a7, 0, 0, 0);
         }
      }
      const double r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
      if (r == 12345.678) out[0] = r;
      return;
   }
   const bool do_mfma = MODE == 0 || (MODE == 2 && (wave & 4) == 0);   // waves 0-3 -> one per SIMD; 4-7 their partners
   double r = 0;
   if (do_mfma) {
      v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      double x = seed + threadIdx.x * 1e-9, y = 1.0 - seed;
      for (int i = 0; i < iters; i++) {
#pragma unroll
         for (int u = 0; u < 8; u++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
         }
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
   }
   else {
      double c[16];
#pragma unroll
      for (int j = 0; j < 16; j++) c[j] = seed * j;
      double x = seed + threadIdx.x * 1e-9, y = 1.0 - seed;
      for (int i = 0; i < iters; i++) {
#pragma unroll
         for (int u = 0; u < 8; u++)
#pragma unroll
            for (int j = 0; j < 16; j++) c[j] = fma(x, c[j], y);
      }
#pragma unroll
      for (int j = 0; j < 16; j++) r += c[j];
   }
   if (r == 12345.678) out[0] = r;
}

template <int MODE>
double run(int blocks, int threads, int iters, double *d, const char *name)
{
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10, 0.5);
   hipDeviceSynchronize();
   hipEventRecord(e0);
   hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.5);
   hipEventRecord(e1);
   hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1);
   const double waves = (double)blocks * threads / 64;
   double mf = 0, vf = 0;
   if (MODE == 0) mf = waves * iters * 32.0 * 2048;
   if (MODE == 1) vf = waves * iters * 128.0 * 64 * 2;
   if (MODE == 2) { mf = waves / 2 * iters * 32.0 * 2048; vf = waves / 2 * iters * 128.0 * 64 * 2; }
   if (MODE == 3) mf = waves * iters * 32.0 * 512;
   double h[3] = {0, 0, 0};
   hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
   printf("%-34s blocks=%d thr=%d  %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF  shader clock %.0f MHz\n", name, blocks, threads, ms,
          mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9, h[2] > 0 ? h[1] / h[2] * 100.0 : 0.0);
   return ms;
}

int main()
{
   double *d; hipMalloc(&d, 64);
   hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
   printf("device %s  CUs %d  clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
   const int cu = p.multiProcessorCount;
   run<0>(cu, 256, 4000, d, "mfma f64 16x16x4, 1 wave/SIMD");
   run<0>(cu, 512, 4000, d, "mfma f64 16x16x4, 2 waves/SIMD");
   run<0>(cu * 2, 512, 4000, d, "mfma f64 16x16x4, 4 waves/SIMD");
   run<1>(cu, 256, 4000, d, "v_fma_f64, 1 wave/SIMD");
   run<1>(cu, 512, 4000, d, "v_fma_f64, 2 waves/SIMD");
   run<1>(cu * 2, 512, 4000, d, "v_fma_f64, 4 waves/SIMD");
   run<3>(cu, 256, 4000, d, "mfma f64 4x4x4 (4 blocks), 1 wave/SIMD");
   run<3>(cu * 2, 512, 4000, d, "mfma f64 4x4x4 (4 blocks), 4 waves/SIMD");
   run<2>(cu, 512, 4000, d, "mixed: 1 mfma + 1 valu wave/SIMD");
   run<2>(cu * 2, 512, 4000, d, "mixed: 2 mfma + 2 valu waves/SIMD");
   return 0;
}
