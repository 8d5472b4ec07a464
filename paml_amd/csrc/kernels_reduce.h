// kernels_reduce.h — Kernel C: per-pattern class mixture + log and the deterministic two-level weighted sum.
#pragma once
#include <hip/hip_runtime.h>

#include "kernel_args.h"

namespace paml_amd {

// ------------------------------------------------------------------------------------------------
// Reduction: per-pattern class mixture + log (lfundG treesub.c:7630-7657, lfun 7796-7800), then a
// fixed-order two-level sum of w_h * log f_h (deterministic for a given n_patt).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pattern_lnf(const ReduceArgs &a, int h)
{
   if (a.mode == PAML_AMD_MODE_LFUN) return a.fhK[h];
   double fh;
   if (a.n_scale) {
      int it = 0;
      for (int ir = 1; ir < a.K; ir++)
         if (a.fhK[(long)ir * a.n_patt + h] > a.fhK[(long)it * a.n_patt + h]) it = ir;
      const double t = a.fhK[(long)it * a.n_patt + h];
      fh = 0;
      for (int ir = 0; ir < a.K; ir++) fh += a.freqK[ir] * exp(a.fhK[(long)ir * a.n_patt + h] - t);
      return t + log(fh);
   }
   fh = 0;
   for (int ir = 0; ir < a.K; ir++) fh += a.freqK[ir] * a.fhK[(long)ir * a.n_patt + h];
   if (fh <= 0) fh = 1e-300;
   return log(fh);
}

__global__ __launch_bounds__(256) void reduce_stage1(ReduceArgs a)
{
   if (blockIdx.y) {
      const long off = (long)blockIdx.y * a.K * a.n_patt;
      a.fhK += off;
      if (a.fscale) a.fscale += off;
      a.freqK += blockIdx.y * a.freqK_bs;
      a.partial += (long)blockIdx.y * a.nb_stride;
      if (a.lnf) a.lnf += (long)blockIdx.y * a.n_patt;
   }
   const int lo = blockIdx.x * a.chunk;
   const int hi = min(a.n_patt, lo + a.chunk);
   double acc = 0;
   for (int h = lo + threadIdx.x; h < hi; h += 256) {
      double v = 0;
      if (a.weights[h] > 0) {
         if (a.raw && (a.mode == PAML_AMD_MODE_LFUN || a.n_scale)) {   // the log the specialised kernel leaves to us
            for (int ir = 0; ir < a.K; ir++) {
               const long ix = (long)ir * a.n_patt + h;
               a.fhK[ix] = log(a.fhK[ix]) + (a.n_scale ? a.fscale[ix] : 0.0);
            }
         }
         v = pattern_lnf(a, h);
         acc += v * a.weights[h];
      }
      if (a.lnf) a.lnf[h] = v;
   }
   red_block_finish(acc, a.partial, a.first_chunk + blockIdx.x, a.nb_stride, a.out + blockIdx.y, a.counter ? a.counter + blockIdx.y * RED_TICKET_WORDS : nullptr);
}

__global__ __launch_bounds__(256) void reduce_stage2(const double *partial, int nb, double *out)
{
   __shared__ double sw[4];
   double acc = 0;
   partial += (long)blockIdx.x * nb;      // one block per batch element
   out += blockIdx.x;
   for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
#pragma unroll
   for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
   if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
   __syncthreads();
   if (threadIdx.x == 0) *out = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

}  // namespace paml_amd
