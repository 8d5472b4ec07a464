// kernels_branch.h — Kernel D: branch-local lnL(t), dlnL, ddlnL on resident partials, and the node posteriors.
#pragma once
#include <hip/hip_runtime.h>

#include "kernel_args.h"

namespace paml_amd {

// ------------------------------------------------------------------------------------------------
// Branch-local evaluation (lfuntdd / lfuntdd_SiteClass, treesub.c:8204-8296, 8403-8541).
// pmat_deriv_kernel: P, dP, ddP = sum_k U[:,k] e^{t mu_k} {1, mu_k, mu_k^2} V[k,:] (plain exp, k = 0 term forced
// to 1, no clamp; Cijk form for baseml), mu_k = rgene * Root_k * rateSite * Qfactor, for every trial length,
// gene and class.  branch_kernel: one pattern per lane,
//   f = sum_ir freqK_ir sum_{i in B} pi_i B_i sum_j P_ij A_j   (and f', f'' with dP, ddP)
// from the two partials across the branch (A exported by the pruning kernel run on the re-rooted tree,
// B = exported partial of the lower node or the state set of a tip), then the weighted sums of
// log f, f'/f and (f f'' - f'^2)/f^2 with a fixed-order two-level reduction.
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void pmat_deriv_kernel(DerivArgs a)
{
   const int it = blockIdx.x, pset = blockIdx.y, n = a.n;
   const int gene = pset / a.K, iclass = pset % a.K;
   const EigenDev es = a.eigen[a.eigen_of[(gene * a.K + iclass) * a.n_labels + a.label]];
   const double t = a.t[it];
   const double qf = es.kind == PAML_AMD_EIGEN_UVROOT ? a.qfactor[iclass * a.n_labels + a.label] : 1.0;
   const double base = a.gene_rate[gene] * a.rate[gene * a.rate_gs + iclass] * qf;
   const int nroot = es.kind == PAML_AMD_EIGEN_CIJK ? es.nR : n;
   double *P = a.out + ((long)(pset * a.n_t + it) * 3) * n * n, *dP = P + n * n, *ddP = dP + n * n;
   __shared__ double sE[64], sM[64];
   if (es.kind == PAML_AMD_EIGEN_K80 || es.kind == PAML_AMD_EIGEN_JC69LIKE) {
      // closed forms (PMatK80 tools.c:578, PMatJC69like codeml.c:3585) written with their two / one non-zero rates:
      //   K80: mu1 = -4/(kappa+2) (all changes), mu2 = -2(kappa+1)/(kappa+2) (within transitions);  JC-like: mu = -n/(n-1)
      const bool k80 = es.kind == PAML_AMD_EIGEN_K80;
      const double m1 = base * (k80 ? -4 / (es.kappa + 2) : -(double)n / (n - 1)), m2 = base * (k80 ? -2 * (es.kappa + 1) / (es.kappa + 2) : 0.0);
      const double e1 = exp(t * m1), e2 = k80 ? exp(t * m2) : 0.0;
      for (int idx = threadIdx.x; idx < n * n; idx += 256) {
         const int i = idx / n, j = idx % n;
         double c1, c2;      // P = 1/n + c1 e1 + c2 e2
         if (k80) { c1 = (i == j || (i ^ j) == 1) ? 0.25 : -0.25; c2 = i == j ? 0.5 : ((i ^ j) == 1 ? -0.5 : 0.0); }
         else { c1 = i == j ? 1 - 1.0 / n : -1.0 / n; c2 = 0; }
         P[idx] = 1.0 / n + c1 * e1 + c2 * e2;
         dP[idx] = c1 * e1 * m1 + c2 * e2 * m2;
         ddP[idx] = c1 * e1 * m1 * m1 + c2 * e2 * m2 * m2;
      }
   }
   else {
      for (int k = threadIdx.x; k < nroot; k += 256) {
         const double mu = base * es.Root[k];     // treesub.c:8479: rgene * Root[k] * _rateSite (* Qfactor)
         sM[k] = mu;
         sE[k] = k ? exp(t * mu) : 1.0;
      }
      __syncthreads();
      for (int idx = threadIdx.x; idx < n * n; idx += 256) {
         const int i = idx / n, j = idx % n;
         double p = 0, dp = 0, ddp = 0;
         for (int k = 0; k < nroot; k++) {
            const double c0 = es.kind == PAML_AMD_EIGEN_CIJK ? es.Cijk[((long)i * n + j) * nroot + k] * sE[k]
                                                             : (es.U[i * n + k] * sE[k]) * es.V[k * n + j];
            p += c0;
            if (k) {
               dp += c0 * sM[k];
               ddp += c0 * sM[k] * sM[k];
            }
         }
         P[idx] = p; dP[idx] = dp; ddP[idx] = ddp;
      }
   }
   if (a.frag) {      // element ((kb2*4 + jb)*64 + lane)*2 + e  =  M[jb*16 + (lane&15)][4*(2*kb2+e) + (lane>>4)], zero padded
      __syncthreads();
      __threadfence_block();
      for (int d = 0; d < 3; d++) {
         const double *M = P + (long)d * n * n;
         double *pf = a.frag + ((long)(pset * a.n_t + it) * 3 + d) * 4096;
         for (int idx = threadIdx.x; idx < 4096; idx += 256) {
            const int e = idx & 1, lane = (idx >> 1) & 63, jb = (idx >> 7) & 3, kb2 = idx >> 9;
            const int r = jb * 16 + (lane & 15), c = 4 * (2 * kb2 + e) + (lane >> 4);
            pf[idx] = (r < n && c < n) ? M[r * n + c] : 0.0;
         }
      }
   }
}


__global__ __launch_bounds__(256) void branch_kernel(BranchArgs a)
{
   __shared__ double sw[4][3];
   const int n = a.n, h = blockIdx.x * 256 + threadIdx.x;
   const bool valid = h < a.n_patt && a.weights[h < a.n_patt ? h : 0] > 0;
   int gene = 0;
   if (valid)
      while (gene + 1 < a.n_genes && h >= a.gene_off[gene + 1]) gene++;
   const double *pi = a.pi + (long)(a.n_pi > 1 ? gene : 0) * n;
   for (int it = 0; it < a.n_t; it++) {
      double fh = 0, dfh = 0, ddfh = 0, smax = 0;
      if (valid) {
         // with scaling nodes class ir's sums carry the factor exp(S_ir): bring the classes to the common factor
         // exp(smax) (lfuntdd_SiteClass treesub.c:8316-8332 does the same with its own pivot)
         if (a.SA) {
            smax = -1e300;
            for (int ir = 0; ir < a.K; ir++) {
               double s = 0;
               for (int k = 0; k < a.n_scale; k++) s += a.SA[((long)ir * a.n_scale + k) * a.n_patt + h];
               smax = s > smax ? s : smax;
            }
         }
         for (int ir = 0; ir < a.K; ir++) {
            double cs = 1.0;
            if (a.SA) {
               double s = 0;
               for (int k = 0; k < a.n_scale; k++) s += a.SA[((long)ir * a.n_scale + k) * a.n_patt + h];
               cs = exp(s - smax);
            }
            const double *Ah = a.A + (long)ir * a.cls_stride + (long)h * n;
            const double *M = a.PdP + ((long)((gene * a.K + ir) * a.n_t + it) * 3) * n * n;
            const int code = a.b_is_tip ? a.zb[h] : 0;
            const int n1 = a.b_is_tip ? a.n_chara[code] : n;
            for (int ii = 0; ii < n1; ii++) {
               const int i = a.b_is_tip ? a.chara_map[code * n + ii] : ii;
               const double bi = a.b_is_tip ? 1.0 : a.B[(long)ir * a.cls_stride + (long)h * n + i];
               const double piqi = a.freqK[ir] * pi[i] * bi * cs;
               double pq = 0, dpq = 0, ddpq = 0;
               const double *Pi = M + (long)i * n, *dPi = Pi + n * n, *ddPi = dPi + n * n;
               for (int j = 0; j < n; j++) {
                  const double aj = Ah[j];
                  pq += Pi[j] * aj;
                  dpq += dPi[j] * aj;
                  ddpq += ddPi[j] * aj;
               }
               fh += piqi * pq;
               dfh += piqi * dpq;
               ddfh += piqi * ddpq;
            }
         }
      }
      double v0 = 0, v1 = 0, v2 = 0;
      if (valid) {
         const double w = a.weights[h];
         v0 = (log(fh) + smax) * w;
         v1 = dfh / fh * w;
         v2 = (fh * ddfh - dfh * dfh) / (fh * fh) * w;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
         v0 += __shfl_xor(v0, off);
         v1 += __shfl_xor(v1, off);
         v2 += __shfl_xor(v2, off);
      }
      __syncthreads();
      if ((threadIdx.x & 63) == 0) {
         sw[threadIdx.x >> 6][0] = v0; sw[threadIdx.x >> 6][1] = v1; sw[threadIdx.x >> 6][2] = v2;
      }
      __syncthreads();
      if (threadIdx.x < 3)
         a.partial[((long)blockIdx.x * a.n_t + it) * 3 + threadIdx.x] =
            (sw[0][threadIdx.x] + sw[1][threadIdx.x]) + (sw[2][threadIdx.x] + sw[3][threadIdx.x]);
   }
}

// ---- the same contraction for the 21..64-state engines, on the matrix cores ---------------------------------------------------
// The two partials across the branch are resident in the pruning kernels' own layout ([class][node][16-pattern group][m][lane],
// OP_STORE), so a wave reads its 16 patterns' A and B as sixteen coalesced 512-byte loads each; P, dP and ddP arrive in MFMA
// A-operand order (pmat_deriv_kernel's frag output), are staged through LDS by LDS-DMA exactly as the pruning kernel stages a
// branch's P, and y = M . A is the pruning kernel's 64-MFMA matvec.  f, f', f'' = sum_i pi_i B_i y_i: sixteen FMAs per lane and
// two cross-lane adds.  One launch per trial length; all classes inside (their mixture is per pattern).

__global__ __launch_bounds__(256, 2) void branch_mfma_kernel(BranchMfmaArgs a)
{
   constexpr int WAVES = 4;
   __shared__ __attribute__((aligned(16))) double sP[2][4096];
   __shared__ double sw[4][3];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, hl = lane & 15;
   const int tile = blockIdx.x;
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y;
   const int hend = as_const(a.gene_off)[gene + 1];
   const int h = h0 + wave * 16 + hl;
   const bool valid = h < hend;
   const int hc = valid ? h : hend - 1;
   const double *pq = a.pi + (long)(a.n_pi > 1 ? gene : 0) * 64 + q * 16;
   const long groups = (long)a.n_tiles * WAVES, grp = (long)tile * WAVES + wave;
   const bool b_tip = a.b_node < a.n_tips;

   // first matrix in flight while the scale factors are read
   const double *frag0 = a.frag + (((long)gene * a.K * a.n_t + a.it) * 3) * 4096;      // class 0, derivative 0
   stage_p<WAVES>(frag0, sP[0], wave, lane);
   double smax = 0;
   if (a.scalef) {
      smax = -1e300;
      for (int ir = 0; ir < a.K; ir++) {
         double s = 0;
         for (int k = 0; k < a.n_scale; k++) s += a.scalef[((long)ir * a.n_scale + k) * a.n_patt + hc];
         smax = s > smax ? s : smax;
      }
   }
   double f[3] = {0, 0, 0};
   int buf = 0;
   for (int ir = 0; ir < a.K; ir++) {
      double cur[16], bv[16];
      const double *pa = a.partials + (((long)ir * a.n_int + (a.a_node - a.n_tips)) * groups + grp) * 1024;
#pragma unroll
      for (int m = 0; m < 16; m++) cur[m] = pa[m * 64 + lane];
      if (b_tip) {
         const unsigned long long mask = a.code_mask[a.zb[hc]];
#pragma unroll
         for (int m = 0; m < 16; m++) bv[m] = ((mask >> (4 * m + q)) & 1ull) ? 1.0 : 0.0;
      }
      else {
         const double *pb = a.partials + (((long)ir * a.n_int + (a.b_node - a.n_tips)) * groups + grp) * 1024;
#pragma unroll
         for (int m = 0; m < 16; m++) bv[m] = pb[m * 64 + lane];
      }
#pragma unroll
      for (int m = 0; m < 16; m++) bv[m] *= pq[m];
      double cs = 1.0;
      if (a.scalef) {
         double s = 0;
         for (int k = 0; k < a.n_scale; k++) s += a.scalef[((long)ir * a.n_scale + k) * a.n_patt + hc];
         cs = exp(s - smax);
      }
      const double wgt = a.freqK[ir] * cs;
      for (int d = 0; d < 3; d++) {
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         __syncthreads();      // this matrix has landed in sP[buf]; every wave is done with sP[buf ^ 1]
         const int nd = d == 2 ? 0 : d + 1, nir = d == 2 ? ir + 1 : ir;
         if (nir < a.K)
            stage_p<WAVES>(a.frag + ((((long)gene * a.K + nir) * a.n_t + a.it) * 3 + nd) * 4096, sP[buf ^ 1], wave, lane);
         v4d acc[4];
         mfma_matvec(sP[buf], lane, cur, acc);
         double g = 0;
#pragma unroll
         for (int m = 0; m < 16; m++) g = fma(bv[m], acc[m >> 2][m & 3], g);
         g += __shfl_xor(g, 16);
         g += __shfl_xor(g, 32);
         f[d] = fma(wgt, g, f[d]);
         buf ^= 1;
      }
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   double v0 = 0, v1 = 0, v2 = 0;
   if (valid && q == 0 && a.weights[hc] > 0) {
      const double w = a.weights[hc];
      v0 = (log(f[0]) + smax) * w;
      v1 = f[1] / f[0] * w;
      v2 = (f[0] * f[2] - f[1] * f[1]) / (f[0] * f[0]) * w;
   }
#pragma unroll
   for (int off = 32; off >= 1; off >>= 1) {
      v0 += __shfl_xor(v0, off);
      v1 += __shfl_xor(v1, off);
      v2 += __shfl_xor(v2, off);
   }
   if (lane == 0) { sw[wave][0] = v0; sw[wave][1] = v1; sw[wave][2] = v2; }
   __syncthreads();
   if (tid < 3) a.partial[((long)tile * a.n_t + a.it) * 3 + tid] = (sw[0][tid] + sw[1][tid]) + (sw[2][tid] + sw[3][tid]);
}

__global__ __launch_bounds__(256) void branch_reduce_kernel(const double *partial, int nb, int n_out, double *out)
{
   __shared__ double sw[4];
   for (int o = 0; o < n_out; o++) {
      double acc = 0;
      for (int i = threadIdx.x; i < nb; i += 256) acc += partial[(long)i * n_out + o];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
      __syncthreads();
      if (threadIdx.x == 0) out[o] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
   }
}

// ------------------------------------------------------------------------------------------------
// Marginal posterior of the states at a node (PostProbNode treesub.c:6142, AncestralMarginal 6288): with the tree rooted
// at the node, L[k][h][i] = exported partial of class k; post[h][i] = sum_k freqK_k pi_i L[k][h][i] e^{S_k} / sum_i(...).
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void posterior_kernel(PostArgs a)
{
   const int h = blockIdx.x * 256 + threadIdx.x, n = a.n;
   if (h >= a.n_patt) return;
   int gene = 0;
   while (gene + 1 < a.n_genes && h >= a.gene_off[gene + 1]) gene++;
   const double *pi = a.pi + (long)(a.n_pi > 1 ? gene : 0) * n;
   double smax = 0;
   if (a.S) {
      smax = -1e300;
      for (int k = 0; k < a.K; k++) smax = fmax(smax, a.S[(long)k * a.n_patt + h]);
   }
   double tot = 0;
   for (int i = 0; i < n; i++) {
      double v = 0;
      for (int k = 0; k < a.K; k++) {
         const double cs = a.S ? exp(a.S[(long)k * a.n_patt + h] - smax) : 1.0;
         v += a.freqK[k] * cs * a.L[((long)k * a.n_patt + h) * n + i];
      }
      v *= pi[i];
      a.post[(long)h * n + i] = v;
      tot += v;
   }
   const double inv = tot > 0 ? 1.0 / tot : 0.0;
   for (int i = 0; i < n; i++) a.post[(long)h * n + i] *= inv;
}

}  // namespace paml_amd
