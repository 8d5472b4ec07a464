// tools/mfma_f64_mix.hip — does f64 MFMA share the SIMD's VALU issue with a co-resident wave?
// even waves: back-to-back v_mfma_f64_16x16x4_f64; odd waves: one of {idle, int VALU, f32 VALU, v_mov, LDS reads,
// f64 mul}.  Reports the MFMA rate next to the partner's instruction rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int PARTNER>
__global__ __launch_bounds__(512) void k(double *out, int iters, double seed, unsigned long long *cnt)
{
   __shared__ double lds[4096];
   const int wave = threadIdx.x >> 6;
   for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = seed * i;
   __syncthreads();
   double r = 0;
   if ((wave & 4) == 0) {   // waves 0-3 land on SIMDs 0,2,1,3 (one each); waves 4-7 are their partners
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
   else if (PARTNER != 0) {
      // the partner spins until the MFMA waves are done would be unfair; instead it runs a fixed amount of
      // work (iters * 256 instructions) and we report how the MFMA time changes
      unsigned v[8];
      float f[8];
      double d[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { v[j] = threadIdx.x + j; f[j] = seed * j; d[j] = seed * j; }
      for (int i = 0; i < iters; i++) {
#pragma unroll
         for (int u = 0; u < 32; u++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
               if (PARTNER == 1) v[j] = v[j] * 3u + 1u;                 // int VALU (v_mad_u32_u24 / v_mul_lo)
               if (PARTNER == 2) f[j] = fmaf(f[j], 1.0001f, 0.5f);      // f32 FMA
               if (PARTNER == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 1) & 7]));
               if (PARTNER == 4) d[j] = d[j] * 1.0000001;               // v_mul_f64
               if (PARTNER == 5) d[j] += lds[(threadIdx.x * 2 + j * 128 + u) & 4095];   // LDS read + f64 add
               if (PARTNER == 6) asm volatile("v_add_u32 %0, %1, %2" : "=v"(v[j]) : "v"(v[j]), "v"(v[(j + 1) & 7]));
            }
         }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) r += v[j] + f[j] + d[j];
   }
   if (r == 12345.678) out[0] = r;
   (void)cnt;
}

template <int PARTNER>
void run(int blocks, int iters, double *d, const char *name)
{
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   hipLaunchKernelGGL(k<PARTNER>, dim3(blocks), dim3(512), 0, 0, d, 10, 0.5, nullptr);
   hipDeviceSynchronize();
   hipEventRecord(e0);
   hipLaunchKernelGGL(k<PARTNER>, dim3(blocks), dim3(512), 0, 0, d, iters, 0.5, nullptr);
   hipEventRecord(e1);
   hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1);
   const double mf = (double)blocks * 4 * iters * 32.0 * 2048;     // 4 MFMA waves per block
   const double pi = (double)blocks * 4 * iters * 256.0;            // partner wave-instructions
   printf("%-28s %.3f ms  mfma-if-alone %.1f TF-equivalent  partner %.2f Ginstr/s (wave-instr)\n", name, ms, mf / ms / 1e9,
          pi / ms / 1e6);
}

int main()
{
   double *d; hipMalloc(&d, 64);
   const int cu = 256, it = 2000;
   run<0>(cu, it, d, "mfma + idle partner");
   run<1>(cu, it, d, "mfma + int mul-add");
   run<2>(cu, it, d, "mfma + f32 fma");
   run<3>(cu, it, d, "mfma + v_mov_b32");
   run<6>(cu, it, d, "mfma + v_add_u32");
   run<4>(cu, it, d, "mfma + v_mul_f64");
   run<5>(cu, it, d, "mfma + lds read/f64 add");
   return 0;
}
