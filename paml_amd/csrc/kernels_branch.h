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
      part_load(pa, lane, cur);
      if (b_tip) {
         const unsigned long long mask = a.code_mask[a.zb[hc]];
#pragma unroll
         for (int m = 0; m < 16; m++) bv[m] = ((mask >> (4 * m + q)) & 1ull) ? 1.0 : 0.0;
      }
      else {
         const double *pb = a.partials + (((long)ir * a.n_int + (a.b_node - a.n_tips)) * groups + grp) * 1024;
         part_load(pb, lane, bv);
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

// ---- the contraction in the eigen basis (21..64 states, one gene, (U, V, Root) eigen systems) -------------------------------------
// lfuntdd's f_h(t) = sum_i pi_i B_i sum_j P_ij(t) A_j with P(t) = U diag(e^{mu_k t}) V is  sum_k e^{mu_k t} z_k w_k  with
// w = V A and z = U^T (pi o B): two matrix products per pattern for ANY number of trial lengths and for f, f', f'' alike (the
// reference's form, and round 2's kernel, spend three per trial length: P, dP, ddP), and the coefficients c_k = z_k w_k — one
// 64-vector per pattern and class, kept in HBM beside the partials — serve every further trial length on the same branch with
// no matrix product at all (minbranches asks for l, l', l'' at t0 and then for up to four trial lengths per Newton step, on the
// same branch: treesub.c:8039-8117).  B a tip: z is a table row of the tip's code, one product per pattern.
//
// branch_eigprep_kernel   per class: V and U^T diag(pi) in MFMA A-operand order, the tips' z rows, e^{mu t} {1, mu, mu^2} per trial length
// branch_eig_kernel       persistent workgroups (8 waves, one class per blockIdx.y, its <= 4 matrices resident in LDS for the launch);
//                         a wave owns 16 patterns at a time: A (read, or formed from its sons' partials / tip rows and written back —
//                         the one node whose orientation changes when minbranches moves to the next branch), w = V A, z, c -> HBM;
//                         with one class also f, f', f'' of up to BEIG_NT trial lengths and the chunk's three sums per trial length
// branch_poly_kernel      f, f', f'' from the stored coefficients: the class mixture, more than BEIG_NT trial lengths, every later call
// branch_total_kernel     fixed-order total of the rows (after the all-reduce over the ranks, when there are ranks)
// A row of `partial` ([column][row]) is the sum over one wave's eighth of a reduction chunk of the evaluation (a function of the GLOBAL
// pattern count: the same rows whatever the number of ranks), formed in one fixed order (the wave's patterns in sequence, then the
// sixteen pattern columns by butterfly); no workgroup barrier anywhere in the walk.
__global__ __launch_bounds__(256) void branch_eigprep_kernel(EigPrepArgs a)
{
   const int iclass = blockIdx.x, n = a.n, tid = threadIdx.x;
   const EigenDev es = a.eigen[a.eigen_of[iclass * a.n_labels + a.label]];
   const double base = a.gene_rate[0] * a.rate[iclass] * a.qfactor[iclass * a.n_labels + a.label];      // treesub.c:8479: rgene * _rateSite * Qfactor (x Root[k])
   double *et = a.etab + (long)iclass * a.n_t * 192;
   for (int idx = tid; idx < a.n_t * 64; idx += 256) {
      const int it = idx >> 6, el = idx & 63, k = 4 * (el & 15) + (el >> 4);
      double e0 = 0, e1 = 0, e2 = 0;
      if (k == 0) e0 = 1.0;      // (the k = 0 term is exp(0) and drops out of the derivatives, as in lfuntdd)
      else if (k < n) {
         const double mu = base * es.Root[k];
         e0 = exp(a.t[it] * mu); e1 = e0 * mu; e2 = e1 * mu;
      }
      et[(it * 3 + 0) * 64 + el] = e0; et[(it * 3 + 1) * 64 + el] = e1; et[(it * 3 + 2) * 64 + el] = e2;
   }
   if (a.only_etab) return;
   // element ((kb2*4 + jb)*64 + lane)*2 + e  =  M[jb*16 + (lane&15)][4*(2*kb2+e) + (lane>>4)], zero padded (pmat_kernel's order)
   double *fv = a.efrag + (long)iclass * 2 * 4096, *fu = fv + 4096;
   for (int idx = tid; idx < 4096; idx += 256) {
      const int e = idx & 1, lane = (idx >> 1) & 63, jb = (idx >> 7) & 3, kb2 = idx >> 9;
      const int r = jb * 16 + (lane & 15), c = 4 * (2 * kb2 + e) + (lane >> 4);
      const bool in = r < n && c < n;
      fv[idx] = in ? es.V[r * n + c] : 0.0;                    // w_k = sum_j V[k][j] A_j
      fu[idx] = in ? es.U[c * n + r] * a.pi[c] : 0.0;          // z_k = sum_i U[i][k] pi_i B_i
   }
   if (a.ecol && tid < 128) {      // 61 states: column 60 of both matrices as the lanes' accumulators want it (rank-1 tail of the product, jit_col_seed)
      const int k = 4 * (tid & 15) + ((tid & 63) >> 4);
      a.ecol[(long)iclass * 128 + tid] = k < n ? (tid < 64 ? es.V[k * n + 60] : es.U[60 * n + k] * a.pi[60]) : 0.0;
   }
   double *zt = a.ztab + (long)iclass * a.n_codes * 64;
   for (int idx = tid; idx < a.n_codes * 64; idx += 256) {
      const int code = idx >> 6, el = idx & 63, k = 4 * (el & 15) + (el >> 4);
      const unsigned long long mask = a.code_mask[code];
      double s = 0;
      if (k < n)
         for (int i = 0; i < n; i++)
            if ((mask >> i) & 1ull) s += es.U[i * n + k] * a.pi[i];
      zt[idx] = s;
   }
}

// The partials and coefficients of a call are streamed through once (0.5 GB each at 16 taxa x 10^6 patterns) and nothing of them is
// read again before the caches have turned over: non-temporal accesses.  BEIG_STREAM (tools/build_variant.sh): bit 0 = the stores,
// bit 1 = the loads.  Measured on MI355X (gpurun_out/r04b/branch_nt.txt -> profiles/r04_branch.txt): the coefficient-forming kernel
// is indifferent (0.445 / 0.443 / 0.447 / 0.444 ms per call for 0 / 1 / 2 / 3), the polynomial kernel on the stored coefficients
// gains 8-15 % from the loads (a further trial length 0.198 -> 0.182 ms, four 0.168 -> 0.150, in the walk 0.215 -> 0.183).
#ifndef BEIG_STREAM
#define BEIG_STREAM 3
#endif
__device__ __forceinline__ void beig_load(const double *p, int lane, v4d (&x)[4])      // (part_load's layout, straight into the MFMA tuples)
{
   const part2_t *p2 = (const part2_t *)p + lane;
#pragma unroll
   for (int i = 0; i < 8; i++) {
#if BEIG_STREAM & 2
      const part2_t v = __builtin_nontemporal_load(p2 + i * 64);
#else
      const part2_t v = p2[i * 64];
#endif
      x[i >> 1][(2 * i) & 3] = v.x; x[i >> 1][(2 * i + 1) & 3] = v.y;
   }
}
__device__ __forceinline__ void beig_store(double *p, int lane, const v4d (&x)[4])
{
   part2_t *p2 = (part2_t *)p + lane;
#pragma unroll
   for (int i = 0; i < 8; i++) {
      const part2_t v = (part2_t){x[i >> 1][(2 * i) & 3], x[i >> 1][(2 * i + 1) & 3]};
#if BEIG_STREAM & 1
      __builtin_nontemporal_store(v, p2 + i * 64);
#else
      p2[i * 64] = v;
#endif
   }
}
// g[d] += sum_m c_m E_d[k = 4m + q]: the lane's share of f, f', f'' for one trial length (et: [3][64] in LDS, element q*16 + m)
__device__ __forceinline__ void beig_poly(const v4d (&c)[4], const double *et, int q, double (&g)[3])
{
#pragma unroll
   for (int d = 0; d < 3; d++) {
      const v4d *e = (const v4d *)(et + d * 64 + q * 16);
      double s = g[d];
#pragma unroll
      for (int i = 0; i < 4; i++) {
         const v4d ev = e[i];
         s = fma(c[i].x, ev.x, s); s = fma(c[i].y, ev.y, s); s = fma(c[i].z, ev.z, s); s = fma(c[i].w, ev.w, s);
      }
      g[d] = s;
   }
}
// The lanes' shares of f, f', f'' for up to four trial lengths -> lane (q, pattern) ends up with the pattern's totals of trial length
// it = q: a reduce-scatter over the four state-quarter lanes of a pattern (nine exchanges where four all-reduces take twenty-four),
// after which the logarithm and the divisions are done once per pattern and trial length, not by every lane for every trial length.
__device__ __forceinline__ void beig_scatter(const double (&g)[BEIG_NT][3], int q, double (&t)[3])
{
   static_assert(BEIG_NT == 4, "one trial length per state-quarter lane");
   const bool hi = (q & 2) != 0, odd = (q & 1) != 0;
#pragma unroll
   for (int d = 0; d < 3; d++) {
      const double k0 = (hi ? g[2][d] : g[0][d]) + __shfl_xor(hi ? g[0][d] : g[2][d], 32);
      const double k1 = (hi ? g[3][d] : g[1][d]) + __shfl_xor(hi ? g[1][d] : g[3][d], 32);
      t[d] = (odd ? k1 : k0) + __shfl_xor(odd ? k0 : k1, 16);
   }
}
// the pattern's three terms for the lane's trial length: log f (+ the scale factors), f'/f, (f f'' - f'^2)/f^2, weighted (treesub.c:8285-8292)
__device__ __forceinline__ void beig_terms(const double (&t)[3], bool take, double w, double smax, double (&acc)[3])
{
   if (take) {
      acc[0] += (log(t[0]) + smax) * w;
      acc[1] += t[1] / t[0] * w;
      acc[2] += (t[0] * t[2] - t[1] * t[1]) / (t[0] * t[0]) * w;
   }
}
// a row's sums (a wave's run of patterns inside a reduction chunk): over the wave's sixteen pattern columns by butterfly — lane 16 q
// holds trial length q — straight to the row of the partial-sum array: no workgroup barrier anywhere in the walk, the waves of a
// workgroup drift as the matrix-pipe arbitration lets them
// (the array is [column][row]: the total's lanes then read consecutive rows of a column)
__device__ __forceinline__ void beig_row_store(const double (&acc)[3], int nt, int lane, double *row, long n_rows)
{
#pragma unroll
   for (int d = 0; d < 3; d++) {
      double v = acc[d];
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if ((lane & 15) == 0 && (lane >> 4) < nt) row[((lane >> 4) * 3 + d) * n_rows] = v;
   }
}
// the pattern's summed scale factors per class, relative to the largest (lfuntdd_SiteClass treesub.c:8316-8332 uses its own pivot)
__device__ __forceinline__ double beig_smax(const double *scalef, int K, int n_scale, int n_patt, int hc)
{
   double smax = -1e300;
   for (int ir = 0; ir < K; ir++) {
      double s = 0;
      for (int k = 0; k < n_scale; k++) s += scalef[((long)ir * n_scale + k) * n_patt + hc];
      smax = s > smax ? s : smax;
   }
   return smax;
}

#define BEIG_LDS_BYTES ((4 * 4096 + BEIG_NT * 192 + 8 * 3 * BEIG_NT + 4 * 64) * 8)
// NS: sons A's partial is formed from (0: A is resident), S0I / S1I: that son is an internal node (else a tip), BTIP: B is a tip —
// compile-time, so that every instantiation is straight-line code the register allocator can fit into 256 VGPRs without spilling
// T61: 61 states — the sixteenth k-block of every matrix holds the single column 60; its rank-1 term seeds the accumulators on the
// vector pipe and the block's four MFMAs are dropped (60 instead of 64 per product, as in the per-tree pruning kernel)
template <int NS, bool S0I, bool S1I, bool BTIP, bool T61>
__global__ __launch_bounds__(512, 2) void branch_eig_kernel(BranchEigArgs a)
{
   constexpr int WAVES = 8;
   extern __shared__ __attribute__((aligned(16))) double beig_smem[];
   double *sV = beig_smem, *sUt = sV + 4096, *sP0 = sUt + 4096, *sP1 = sP0 + 4096, *sE = sP1 + 4096, *sRed = sE + BEIG_NT * 192,
          *sCol = sRed + 8 * 3 * BEIG_NT;      // [V, U^T, P of son 0, P of son 1][64]
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, hl = lane & 15;
   const int iclass = blockIdx.y;
   constexpr bool b_tip = BTIP, s0_int = NS > 0 && S0I, s1_int = NS > 1 && S1I;
   const double *ef = a.efrag + (long)iclass * 2 * 4096;
   const double *Pint = a.pint + (long)iclass * a.n_nodes * 4096;
   const double *Ptip = a.ptip + (long)iclass * a.n_nodes * a.tip_words;
   stage_p<WAVES>(ef, sV, wave, lane);
   if (!b_tip) stage_p<WAVES>(ef + 4096, sUt, wave, lane);
   if (s0_int) stage_p<WAVES>(Pint + (long)a.son[0] * 4096, sP0, wave, lane);
   if (s1_int) stage_p<WAVES>(Pint + (long)a.son[1] * 4096, sP1, wave, lane);
   if (a.feval)
      for (int idx = tid; idx < a.n_t * 192; idx += WAVES * 64) sE[idx] = a.etab[(long)iclass * a.n_t * 192 + idx];
   if (T61 && tid < 256) {
      const int which = tid >> 6, el = tid & 63;
      double v = 0;
      if (which < 2) v = a.ecol[(long)iclass * 128 + tid];
      else if (which == 2 ? s0_int : s1_int) v = a.pcol[((long)iclass * a.n_nodes + a.son[which - 2]) * 64 + el];
      sCol[tid] = v;
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   __syncthreads();
#define BEIG_MATVEC(MAT, IDX, X, Y)                                                                                             \
   do {                                                                                                                         \
      if (a.no_store & 2) { _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) (Y)[i_] = (X)[i_]; }      /* timing experiment: no products */ \
      else jit_matvec<T61, 4, 16>((MAT), lane, (X), (Y), JitNoSide(), sCol + (IDX) * 64, T61 ? jit_x60((X), lane) : 0.0);       \
   } while (0)

   const long G = a.n_groups;
   double *pclass = a.partials + (long)iclass * a.n_int * G * 1024;
   double *cclass = a.coef + (long)iclass * G * 1024;
   const double *zt = a.ztab + (long)iclass * a.n_codes * 64 + q * 16;
   const double fk = a.freqK[iclass];
   // The operand a group's work starts with — A itself, or the partial of A's first internal son — is requested one group ahead
   // (`pf`), and so are the character codes of the tips involved; the other operands (second son, B, tip rows) at points where a
   // product in front of their use hides their latency, and the two waves of a SIMD overlap the rest.
   const double *first_base = NS == 0 ? pclass + (long)(a.a_node - a.n_tips) * G * 1024
                                      : (s0_int ? pclass + (long)(a.son[0] - a.n_tips) * G * 1024 : nullptr);
   constexpr bool t0 = NS > 0 && !S0I, t1 = NS > 1 && !S1I;      // tip sons
   const unsigned char *z0 = a.z + (long)(t0 ? a.son[0] : 0) * a.n_patt, *z1 = a.z + (long)(t1 ? a.son[1] : 0) * a.n_patt,
                       *zb = a.z + (long)(b_tip ? a.b_node : 0) * a.n_patt;
   // a workgroup takes reduction chunks, each of its eight waves a contiguous eighth of the chunk (a ROW of the partial-sum array)
   const int rg = a.chunk_groups / WAVES;
   auto g_begin = [&](int lc) { return lc * a.chunk_groups + wave * rg; };
   auto g_stop = [&](int lc) { return min(lc * a.chunk_groups + (wave + 1) * rg, a.n_groups); };
   v4d pf[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
   v4d pb[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};      // both partials resident: B is requested a group ahead as well
   constexpr bool pb_ahead = NS == 0 && !b_tip;
   const double *b_base = pclass + (long)((pb_ahead ? a.b_node : a.n_tips) - a.n_tips) * G * 1024;
   int c0 = 0, c1 = 0, cb = 0;
   if ((int)blockIdx.x < a.nb_local && g_begin(blockIdx.x) < g_stop(blockIdx.x)) {
      const int g = g_begin(blockIdx.x), hc = min(g * 16 + hl, a.n_patt - 1);
      if (first_base) beig_load(first_base + (long)g * 1024, lane, pf);
      if (pb_ahead) beig_load(b_base + (long)g * 1024, lane, pb);
      if (t0) c0 = z0[hc];
      if (t1) c1 = z1[hc];
      if (b_tip) cb = zb[hc];
   }
   for (int lc = blockIdx.x; lc < a.nb_local; lc += gridDim.x) {
      double acc[3] = {0, 0, 0};
      const int g_end = g_stop(lc);
      for (int g = g_begin(lc); g < g_end; g++) {
         int gn = g + 1;      // this wave's next group (possibly in the workgroup's next chunk), -1: none
         if (gn >= g_end) {
            const int ln = lc + gridDim.x;
            gn = (ln < a.nb_local && g_begin(ln) < g_stop(ln)) ? g_begin(ln) : -1;
         }
         const int h = g * 16 + hl;
         const bool valid = h < a.n_patt;
         // (no next group: the prefetches re-read this one — unconditional, so that the compiler counts the loads in flight
         //  instead of draining them at a control-flow join)
         const int gp = gn >= 0 ? gn : g;
         const int hc = valid ? h : a.n_patt - 1, hn = min(gp * 16 + hl, a.n_patt - 1);
         const double wt = a.weights[hc];      // (first: a load issued behind the coefficient stores would wait for them)
         int c0n = 0, c1n = 0, cbn = 0;
         if (t0) c0n = z0[hn];
         if (t1) c1n = z1[hn];
         if (b_tip) cbn = zb[hn];
         v4d s1[4], bb[4], w[4], zz[4];
         double2 v1[8];
         if constexpr (NS == 0) {
            if constexpr (b_tip) {
               const v4d *zp = (const v4d *)(zt + (long)cb * 64);
#pragma unroll
               for (int i = 0; i < 4; i++) zz[i] = zp[i];
            }
            BEIG_MATVEC(sV, 0, pf, w);
            beig_load(first_base + (long)gp * 1024, lane, pf);
            if constexpr (pb_ahead) {      // (the second product here, so that its operand's registers take the next group's B at once)
               BEIG_MATVEC(sUt, 1, pb, zz);
               beig_load(b_base + (long)gp * 1024, lane, pb);
            }
         }
         else {
            // A's partial in the tree seen from this branch: the product of its sons' messages (ConditionalPNode, codeml.c:3545-3576)
            v4d x[4];
            if constexpr (s1_int) beig_load(pclass + ((long)(a.son[1] - a.n_tips) * G + g) * 1024, lane, s1);
            if constexpr (t1) tip_gather(Ptip, a.tip_words, a.son[1], c1, q, v1);
            if constexpr (s0_int) {
               BEIG_MATVEC(sP0, 2, pf, x);
               beig_load(first_base + (long)gp * 1024, lane, pf);
            }
            else {
               double2 v[8];
               tip_gather(Ptip, a.tip_words, a.son[0], c0, q, v);
#pragma unroll
               for (int i = 0; i < 8; i++) { x[i >> 1][(2 * i) & 3] = v[i].x; x[i >> 1][(2 * i + 1) & 3] = v[i].y; }
            }
            if constexpr (NS > 1) {
               if constexpr (s1_int) {
                  v4d y[4];
                  BEIG_MATVEC(sP1, 3, s1, y);
#pragma unroll
                  for (int i = 0; i < 4; i++) x[i] = x[i] * y[i];
               }
               else {
#pragma unroll
                  for (int i = 0; i < 8; i++) { x[i >> 1][(2 * i) & 3] *= v1[i].x; x[i >> 1][(2 * i + 1) & 3] *= v1[i].y; }
               }
            }
            if constexpr (b_tip) {
               const v4d *zp = (const v4d *)(zt + (long)cb * 64);
#pragma unroll
               for (int i = 0; i < 4; i++) zz[i] = zp[i];
            }
            else beig_load(pclass + ((long)(a.b_node - a.n_tips) * G + g) * 1024, lane, bb);
            beig_store(pclass + ((long)(a.a_node - a.n_tips) * G + g) * 1024, lane, x);
            BEIG_MATVEC(sV, 0, x, w);
         }
         if constexpr (!b_tip && !pb_ahead) BEIG_MATVEC(sUt, 1, bb, zz);
         double wgt = fk, smax = 0;
         if (a.scalef) {
            smax = beig_smax(a.scalef, a.K, a.n_scale, a.n_patt, hc);
            double s = 0;
            for (int k = 0; k < a.n_scale; k++) s += a.scalef[((long)iclass * a.n_scale + k) * a.n_patt + hc];
            wgt = fk * exp(s - smax);
         }
         v4d c[4];
#pragma unroll
         for (int i = 0; i < 4; i++) c[i] = (w[i] * zz[i]) * wgt;
         if (!(a.no_store & 1)) beig_store(cclass + (long)g * 1024, lane, c);
         if (a.feval) {
            double g4[BEIG_NT][3];
#pragma unroll
            for (int it = 0; it < BEIG_NT; it++) {
               g4[it][0] = g4[it][1] = g4[it][2] = 0;
               if (it < a.n_t) beig_poly(c, sE + it * 192, q, g4[it]);
            }
            double t3[3];
            beig_scatter(g4, q, t3);
            beig_terms(t3, valid && q < a.n_t && wt > 0, wt, smax, acc);
         }
         c0 = c0n; c1 = c1n; cb = cbn;
      }
      if (a.feval) beig_row_store(acc, a.n_t, lane, a.partial + ((long)(a.first_chunk + lc) * WAVES + wave), a.n_rows);
   }
#undef BEIG_MATVEC
}

// dynamic LDS: K x nt_here x 192 doubles of e^{mu t} tables + the reduction scratch
__global__ __launch_bounds__(512) void branch_poly_kernel(BranchPolyArgs a)
{
   constexpr int WAVES = 8;
   extern __shared__ __attribute__((aligned(16))) double bpoly_smem[];
   double *sE = bpoly_smem;
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, hl = lane & 15;
   for (int idx = tid; idx < a.K * a.nt_here * 192; idx += WAVES * 64) {
      const int ir = idx / (a.nt_here * 192), r = idx % (a.nt_here * 192);
      sE[idx] = a.etab[((long)ir * a.n_t + a.it0) * 192 + r];
   }
   __syncthreads();
   const long G = a.n_groups;
   for (int lc = blockIdx.x; lc < a.nb_local; lc += gridDim.x) {
      double acc[3] = {0, 0, 0};
      const int rg = a.chunk_groups / WAVES, g0 = lc * a.chunk_groups + wave * rg, g_end = min(g0 + rg, a.n_groups);
      v4d pf[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};      // the coefficients one (class, group) step ahead
      if (g0 < g_end) beig_load(a.coef + (long)g0 * 1024, lane, pf);
      for (int g = g0; g < g_end; g++) {
         const int h = g * 16 + hl;
         const bool valid = h < a.n_patt;
         const int hc = valid ? h : a.n_patt - 1;
         double gs[BEIG_NT][3];
#pragma unroll
         for (int it = 0; it < BEIG_NT; it++) gs[it][0] = gs[it][1] = gs[it][2] = 0;
         for (int ir = 0; ir < a.K; ir++) {
            v4d c[4];
#pragma unroll
            for (int i = 0; i < 4; i++) c[i] = pf[i];
            if (ir + 1 < a.K) beig_load(a.coef + ((long)(ir + 1) * G + g) * 1024, lane, pf);
            else if (g + 1 < g_end) beig_load(a.coef + (long)(g + 1) * 1024, lane, pf);
#pragma unroll
            for (int it = 0; it < BEIG_NT; it++)
               if (it < a.nt_here) beig_poly(c, sE + ((long)ir * a.nt_here + it) * 192, q, gs[it]);
         }
         const double smax = a.scalef ? beig_smax(a.scalef, a.K, a.n_scale, a.n_patt, hc) : 0.0;
         const double wt = a.weights[hc];
         double t3[3];
         beig_scatter(gs, q, t3);
         beig_terms(t3, valid && q < a.nt_here && wt > 0, wt, smax, acc);
      }
      beig_row_store(acc, a.nt_here, lane, a.partial + ((long)(a.first_chunk + lc) * WAVES + wave) + (long)a.it0 * 3 * a.n_rows, a.n_rows);
   }
}

// one workgroup per output: the fixed-order total of column o of the rows, in reduce_stage2's order (256 lanes sum the rows i, i + 256,
// ... in turn, the butterfly, the four waves pairwise: paml_amd/distributed.py total_fixed_order restates it on the host)
__global__ __launch_bounds__(256) void branch_total_kernel(const double *partial, int nb, int n_out, double *out)
{
   __shared__ double sw[4];
   const int o = blockIdx.x;
   double acc = 0;
   const double *col = partial + (long)o * nb;      // ([column][row])
   for (int i0 = threadIdx.x; i0 < nb; i0 += 256 * 8) {      // eight loads in flight, added in the order i, i + 256, ...
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = i0 + 256 * k < nb ? col[i0 + 256 * k] : 0.0;
#pragma unroll
      for (int k = 0; k < 8; k++)
         if (i0 + 256 * k < nb) acc += v[k];
   }
#pragma unroll
   for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
   if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
   __syncthreads();
   if (threadIdx.x == 0) out[o] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
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
