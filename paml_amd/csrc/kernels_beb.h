// kernels_beb.h — the Bayes-empirical-Bayes grid integral over the class likelihoods of the last evaluation.
#pragma once
#include <hip/hip_runtime.h>

#include "kernel_args.h"

namespace paml_amd {

// ------------------------------------------------------------------------------------------------
// Bayes empirical Bayes grid integral (lfunNSsites_M2M8 codeml.c:6482-6580) over the class likelihoods of the last
// evaluation: n_grid parameter points, each a mixture of n_cls classes (proportion pcl[g][c], class index iw[g][c] into
// the K evaluated classes).  n_grid x n_patt x n_cls terms with a log each — 10^11 at 10^6 patterns.
//   beb_scale:   f[k][h] = fhK[k][h] / max_k fhK[k][h]                       (codeml.c:6297-6305); with scaling nodes fhK holds
//                log f + the scale factors and f[k][h] = exp(fhK[k][h] - max_k fhK[k][h])            (codeml.c:6286-6294)
//   beb_lnfx:    part[g][b] = sum over block b's patterns of w_h log sum_c pcl[g][c] f[iw[g][c]][h]
//   beb_finish:  lnfXs[g] = sum_b part[g][b] (fixed order; with pattern shards all-reduced over the ranks here);  fX = log sum_g exp(lnfXs[g]);  wg[g] = exp(lnfXs[g] - fX)
//   beb_post:    per pattern, sums over the grid of the class posteriors, omega and omega^2
// One pattern per lane with its K class values in registers; the grid tables are wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void beb_scale(BebArgs a)
{
   const int h = blockIdx.x * 256 + threadIdx.x;
   if (h >= a.n_patt) return;
   double mx = a.fhK[h];
   for (int k = 1; k < a.K; k++) mx = fmax(mx, a.fhK[(long)k * a.n_patt + h]);
   if (a.log_form)      // (patterns that do not count have no fhK: f = 0 as in the other branch)
      for (int k = 0; k < a.K; k++) a.f[(long)k * a.n_patt + h] = a.weights[h] > 0 ? exp(a.fhK[(long)k * a.n_patt + h] - mx) : 0.0;
   else
      for (int k = 0; k < a.K; k++) a.f[(long)k * a.n_patt + h] = mx > 0 ? a.fhK[(long)k * a.n_patt + h] / mx : 0.0;
}

__global__ __launch_bounds__(256) void beb_lnfx(BebArgs a)      // grid: (n_pblk, ceil(n_grid / 64))
{
   __shared__ double sw[4];
   const int g0 = blockIdx.y * 64, g1 = min(a.n_grid, g0 + 64);
   const int hlo = blockIdx.x * a.patt_per_blk, hhi = min(a.n_patt, hlo + a.patt_per_blk);
   for (int g = g0; g < g1; g++) {
      const CONST_AS double *pc = as_const(a.pcl + (long)g * a.n_cls);
      const CONST_AS int *ix = as_const(a.iw + (long)g * a.n_cls);
      double acc = 0;
      for (int h = hlo + threadIdx.x; h < hhi; h += 256) {
         const double w = a.weights[h];
         if (!(w > 0)) continue;
         double fh = 0;
         for (int c = 0; c < a.n_cls; c++) fh = fma(pc[c], a.f[(long)ix[c] * a.n_patt + h], fh);
         if (fh >= 1e-300) acc += log(fh) * w;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
      __syncthreads();
      if (threadIdx.x == 0) a.part[(long)g * a.n_pblk + blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
   }
}

__global__ __launch_bounds__(256) void beb_finish(BebArgs a)     // one block
{
   __shared__ double sred[256];
   double mx = -1e300;
   for (int g = threadIdx.x; g < a.n_grid; g += 256) {
      double s;
      if (a.phase == 2) s = a.lnfxs[g];      // (the sums over ALL shards: paml_amd_beb_grid's exchange step)
      else {
         s = 0;
         for (int b = 0; b < a.n_pblk; b++) s += a.part[(long)g * a.n_pblk + b];
         a.lnfxs[g] = s;
      }
      mx = fmax(mx, s);
   }
   if (a.phase == 1) return;
   sred[threadIdx.x] = mx;
   __syncthreads();
   for (int st = 128; st >= 1; st >>= 1) {
      if (threadIdx.x < st) sred[threadIdx.x] = fmax(sred[threadIdx.x], sred[threadIdx.x + st]);
      __syncthreads();
   }
   mx = sred[0];
   __syncthreads();
   double s = 0;
   for (int g = threadIdx.x; g < a.n_grid; g += 256) s += exp(a.lnfxs[g] - mx);
   sred[threadIdx.x] = s;
   __syncthreads();
   for (int st = 128; st >= 1; st >>= 1) {
      if (threadIdx.x < st) sred[threadIdx.x] += sred[threadIdx.x + st];
      __syncthreads();
   }
   const double fx = log(sred[0]) + mx;
   if (threadIdx.x == 0) *a.fx = fx;
   for (int g = threadIdx.x; g < a.n_grid; g += 256) a.wg[g] = exp(a.lnfxs[g] - fx);
}

__global__ __launch_bounds__(256) void beb_post(BebArgs a)
{
   const int h = blockIdx.x * 256 + threadIdx.x;
   const bool valid = h < a.n_patt;
   const int hc = valid ? h : a.n_patt - 1;
   double f[BEB_MAXK];
#pragma unroll
   for (int k = 0; k < BEB_MAXK; k++) f[k] = k < a.K ? a.f[(long)k * a.n_patt + hc] : 0.0;
   double post_last = 0, m1 = 0, m2 = 0;
   for (int g = 0; g < a.n_grid; g++) {
      const CONST_AS double *pc = as_const(a.pcl + (long)g * a.n_cls);
      const CONST_AS int *ix = as_const(a.iw + (long)g * a.n_cls);
      const double wg = as_const(a.wg)[g];
      double fh = 0;
      for (int c = 0; c < a.n_cls; c++) {
         const int k = ix[c];
         double fk = 0;
#pragma unroll
         for (int kk = 0; kk < BEB_MAXK; kk++) fk = (kk == k) ? f[kk] : fk;      // wave-uniform select keeps f in registers
         fh = fma(pc[c], fk, fh);
      }
      if (fh < 1e-300) continue;
      const double inv = wg / fh;
      for (int c = 0; c < a.n_cls; c++) {
         const int k = ix[c];
         double fk = 0;
#pragma unroll
         for (int kk = 0; kk < BEB_MAXK; kk++) fk = (kk == k) ? f[kk] : fk;
         const double t = pc[c] * fk * inv, w = as_const(a.w_class)[k];
         if (c == a.n_cls - 1) post_last += t;
         m1 = fma(t, w, m1);
         m2 = fma(t * w, w, m2);
      }
   }
   if (valid) {
      a.pr_last[h] = post_last;
      a.mean_w[h] = m1;
      const double v = m2 - m1 * m1;
      a.sd_w[h] = v > 0 ? sqrt(v) : 0.0;
   }
}

// Posterior of every mixture class (lfunNSsites_ACD codeml.c:6970-6985: branch-site model A has 4, its 121 evaluated classes
// do not fit the register file, so f is read through L2 — the index is wave-uniform, the access coalesced over patterns):
// post[c][h] = sum_g wg[g] pcl[g][c] f[iw[g][c]][h] / fh(g, h).
__global__ __launch_bounds__(256) void beb_post_classes(BebArgs a)
{
   const int h = blockIdx.x * 256 + threadIdx.x;
   if (h >= a.n_patt) return;
   double post[BEB_MAXCLS], t[BEB_MAXCLS];
#pragma unroll
   for (int c = 0; c < BEB_MAXCLS; c++) post[c] = 0;
   for (int g = 0; g < a.n_grid; g++) {
      const CONST_AS double *pc = as_const(a.pcl + (long)g * a.n_cls);
      const CONST_AS int *ix = as_const(a.iw + (long)g * a.n_cls);
      const double wg = as_const(a.wg)[g];
      double fh = 0;
#pragma unroll
      for (int c = 0; c < BEB_MAXCLS; c++) {
         t[c] = c < a.n_cls ? pc[c] * a.f[(long)ix[c] * a.n_patt + h] : 0.0;
         fh += t[c];
      }
      if (fh < 1e-300) continue;
      const double inv = wg / fh;
#pragma unroll
      for (int c = 0; c < BEB_MAXCLS; c++) post[c] = fma(t[c], inv, post[c]);
   }
   for (int c = 0; c < a.n_cls; c++) a.pr_last[(long)c * a.n_patt + h] = post[c];
}

}  // namespace paml_amd
