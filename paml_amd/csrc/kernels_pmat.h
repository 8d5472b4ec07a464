// kernels_pmat.h — Kernel A: batched construction of the transition-probability matrices (gfx950).  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

#include "kernel_args.h"

namespace paml_amd {

// ------------------------------------------------------------------------------------------------
// Kernel A: batched P(t) — one workgroup per (node, gene x class).
//   UVROOT : P = I + sum_k (U[:,k] expm1(t Root_k)) V[k,:], t<1e-100 -> I, entries <0 -> 0  (tools.c:516-546)
//   CIJK   : P_ij = delta_ij + sum_{k>=1} Cijk[i][j][k] expm1(t Root_k), no clamp           (baseml.c:1572-1589)
//   K80    : closed form (tools.c:578-604);  JC69LIKE: closed form (codeml.c:3585-3595)
// with t = branch * rateSite * rgene [* Qfactor]  (codeml.c:3547-3551, treesub.c:7587).
// Outputs, all for the branch above `node`:
//   rowmajor  [n*n]                P[from*n+to]           (get_pmat; matmul operand of the VALU kernels)
//   frag      [8][4][64][2]        MFMA A-operand order   (mfma64 kernel; see prune_mfma64)
//   tip table [n_codes][...]       column sums over each character code's state set
//                                  (codeml.c:3555-3567), [code][n] for VALU, [code][q][m] for mfma64
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ double pmat_time(const PmatArgs &a, const InlineVec &iv, int bat, int node, int gene, int iclass)
{
   // t = branch * rateSite * rgene (codeml.c:3547-3551)
   const double br = iv.n_branch ? iv.v[node] : a.branch[bat * a.branch_bs + node];
   const double gr = iv.n_branch ? iv.v[iv.n_branch + gene] : a.gene_rate[bat * a.gene_rate_bs + gene];
   return (br * a.rate[bat * a.rate_bs + gene * a.rate_gs + iclass]) * gr;
}

// Models with at most 5 states (the one-pattern-per-lane kernels): 32 threads per matrix, eight matrices per workgroup, no
// 64 x 64 staging — the general kernel below spends 9 us on the 244 4 x 4 matrices of a 32-taxon Gamma-4 evaluation.
// Same arithmetic and summation order as pmat_kernel.  Rate-matrix (UNREST) sets stay with the general kernel.
__global__ __launch_bounds__(256) void pmat_small_kernel(PmatArgs a, InlineVec iv)
{
   __shared__ double sP[8][32];
   const int n = a.n, sub = threadIdx.x >> 5, t5 = threadIdx.x & 31;
   const int KB = a.K * a.B, n_mat = a.n_nodes * a.n_genes * KB;
   const int m = blockIdx.x * 8 + sub;
   const bool on = m < n_mat;
   const int node = on ? m % a.n_nodes : 0, pset = on ? m / a.n_nodes : 0;
   const int gene = pset / KB, bat = (pset % KB) / a.K, iclass = pset % a.K;
   const bool active = on && node != a.root;
   const int i = t5 / n, j = t5 % n;
   const bool ent = active && t5 < n * n;
   double p = 0;
   int lab = 0;
   if (active) lab = a.label[node];
   if (ent) {
      const EigenDev es = a.eigen[a.eigen_of[bat * a.eigen_of_bs + (gene * a.K + iclass) * a.n_labels + lab]];
      double t = pmat_time(a, iv, bat, node, gene, iclass);
      if (es.kind == PAML_AMD_EIGEN_UVROOT) {
         t *= a.qfactor[bat * a.qfactor_bs + iclass * a.n_labels + lab];
         if (t < 1e-100) p = (i == j) ? 1.0 : 0.0;
         else {
            double acc = 0;
            for (int k = 0; k < n; k++) acc = fma(es.U[i * n + k] * expm1(t * es.Root[k]), es.V[k * n + j], acc);
            p = acc + (i == j ? 1.0 : 0.0);
            p = p < 0 ? 0.0 : p;
         }
      }
      else if (es.kind == PAML_AMD_EIGEN_CIJK) {
         const double *c = es.Cijk + ((long)i * n + j) * es.nR;
         double sacc = 0;
         for (int k = 0; k < es.nR; k++) sacc += c[k] * (k >= 1 ? expm1(t * es.Root[k]) : 0.0);
         if (i == j) sacc += 1.0;
         p = sacc;
      }
      else if (es.kind == PAML_AMD_EIGEN_K80) {
         const double kappa = es.kappa;
         const double e1 = expm1(-4 * t / (kappa + 2));
         const bool jc = fabs(kappa - 1) < 1e-20;
         const double e2 = jc ? 0.0 : expm1(-2 * t * (kappa + 1) / (kappa + 2));
         if (jc) p = (i == j) ? 1. + 3 / 4.0 * e1 : -e1 / 4;
         else if (i == j) p = 1 + (e1 + 2 * e2) / 4;
         else if ((i ^ j) == 1) p = (e1 - 2 * e2) / 4;
         else p = -e1 / 4;
      }
      else {   // JC69-like
         const double pii = 1. / n + (1. - 1. / n) * exp(-n / (n - 1.) * t);
         p = i == j ? pii : (1. - pii) / (n - 1.);
      }
   }
   sP[sub][t5] = p;
   __syncthreads();
   if (!active) return;
   const long slot = (long)pset * a.n_nodes + node;
   if (ent) a.rowmajor[slot * n * n + t5] = p;
   if (a.is_leaf[node]) {
      double *pt = a.ptip + slot * a.tip_words;
      for (int idx = t5; idx < a.n_codes * n; idx += 32) {
         const int code = idx / n, jj = idx % n;
         const int nc = a.n_chara[code];
         const unsigned char *map = a.chara_map + code * n;
         double s2 = 0;
         for (int k = 0; k < nc; k++) s2 += sP[sub][jj * n + map[k]];
         pt[idx] = s2;
      }
   }
}

// LD: the side of the padded square the matrices live in while they are built — 64, or 32 for models of at most 32 states that
// do not want the MFMA operand layout (20 states: 25 KB of LDS per workgroup instead of 81, so that six of them share a CU; in a
// run of evaluations P(t) only ever gets the few CUs the persistent pruning kernels leave free).  Same arithmetic per entry.
template <int LD>
__global__ __launch_bounds__(256) void pmat_kernel_t(PmatArgs a, InlineVec iv)
{
   extern __shared__ __attribute__((aligned(16))) double smem[];
   constexpr int ROWS = LD * LD / 256, TL = LD / 4;      // rows per thread of the finished matrix; 4 x 4 register tiles per side
   double *sA = smem;            // [LD][LD]  U*expm1 -> later the finished P (padded with zeros)
   double *sB = smem + LD * LD;  // [LD][LD]  V
   // A workgroup builds the matrices of a.npb consecutive nodes of one parameter set: 1 by default (one node per workgroup, the
   // lowest latency); more is an experiment (engine_eval.hip, PAML_AMD_PMAT_NPB) — U stays in registers and V in LDS while the
   // eigen set does not change, which takes the chain of dependent global loads out of every node but the first.
   const int pset = blockIdx.y, npb = a.npb > 0 ? a.npb : 1;
   const int tid = threadIdx.x, n = a.n;
   const int KB = a.K * a.B;
   const int gene = pset / KB, bat = (pset % KB) / a.K, iclass = pset % a.K;
   // tip branches: the ambiguity map (tools.c:20 nChara / CharaMap) comes to LDS, so that the column-table loop at the end does
   // not chase two dependent global loads per entry
   __shared__ unsigned char sMap[256 * LD];
   __shared__ int sNch[256];
   __shared__ double sE[64];
   bool map_loaded = false;
   int cur = -1, uv_of = -1;      // eigen set held in `es`; ... whose U is in ureg and V in sB
   EigenDev es{};
   double ureg[ROWS];
   const int node_end = min(a.n_nodes, ((int)blockIdx.x + 1) * npb);
   for (int node = blockIdx.x * npb; node < node_end; node++) {
      if (node == a.root) continue;
      const int lab = a.label[node];
      const int ei = a.eigen_of[bat * a.eigen_of_bs + (gene * a.K + iclass) * a.n_labels + lab];
      if (ei != cur) { es = a.eigen[ei]; cur = ei; }
      double t = pmat_time(a, iv, bat, node, gene, iclass);
      const bool leaf = a.is_leaf[node] != 0;
      if (leaf && !map_loaded) {      // (read after the barriers below)
         for (int idx = tid; idx < a.n_codes * n; idx += 256) sMap[idx] = a.chara_map[idx];
         for (int idx = tid; idx < a.n_codes; idx += 256) sNch[idx] = a.n_chara[idx];
         map_loaded = true;
      }

      const int j = tid % LD, rg = tid / LD;   // this thread: column j, rows rg*ROWS .. rg*ROWS+ROWS-1
      double acc[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; r++) acc[r] = 0;

      if (es.kind == PAML_AMD_EIGEN_UVROOT) {
         t *= a.qfactor[bat * a.qfactor_bs + iclass * a.n_labels + lab];
         if (t < 1e-100) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) acc[r] = (rg * ROWS + r == j) ? 1.0 : 0.0;
         }
         else {
            // expm1(t * Root[k]) once per k (not per (i, k))
            if (tid < 64) sE[tid] = tid < n ? expm1(t * es.Root[tid]) : 0.0;
            if (uv_of != cur) {
#pragma unroll
               for (int m = 0; m < ROWS; m++) {
                  const int idx = tid + 256 * m, i = idx / LD, k = idx % LD;
                  const bool in = i < n && k < n;
                  ureg[m] = in ? es.U[i * n + k] : 0.0;
                  sB[idx] = in ? es.V[i * n + k] : 0.0;        // here (i,k) index V as [k'][j'] = [i][k]
               }
               uv_of = cur;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < ROWS; m++) {
               const int idx = tid + 256 * m, i = idx / LD, k = idx % LD;
               sA[k * LD + i] = ureg[m] * sE[k];           // transposed: the four rows of a register tile are contiguous for every k
            }
            __syncthreads();
            {
               // 4 x 4 register tile per thread: four 16-byte LDS reads feed sixteen FMAs; every element still accumulates
               // k ascending (PMatUVRoot's order, tools.c:525-537)
               const int ti = tid / TL, tj = tid % TL;
               const bool tile = tid < TL * TL;
               double c[4][4];
#pragma unroll
               for (int r = 0; r < 4; r++)
#pragma unroll
                  for (int cc = 0; cc < 4; cc++) c[r][cc] = 0;
               for (int k = 0; k < n && tile; k++) {
                  const double2 a0 = *(const double2 *)&sA[k * LD + 4 * ti], a1 = *(const double2 *)&sA[k * LD + 4 * ti + 2];
                  const double2 b0 = *(const double2 *)&sB[k * LD + 4 * tj], b1 = *(const double2 *)&sB[k * LD + 4 * tj + 2];
                  const double av[4] = {a0.x, a0.y, a1.x, a1.y}, bv[4] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
                  for (int r = 0; r < 4; r++)
#pragma unroll
                     for (int cc = 0; cc < 4; cc++) c[r][cc] = fma(av[r], bv[cc], c[r][cc]);
               }
               __syncthreads();
#pragma unroll
               for (int r = 0; r < 4; r++)
#pragma unroll
                  for (int cc = 0; cc < 4; cc++)
                     if (tile) sA[(4 * ti + r) * LD + 4 * tj + cc] = c[r][cc];
               __syncthreads();
#pragma unroll
               for (int r = 0; r < ROWS; r++) acc[r] = sA[(rg * ROWS + r) * LD + j];
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
               int i = rg * ROWS + r;
               double p = acc[r] + (i == j ? 1.0 : 0.0);
               acc[r] = (i < n && j < n) ? (p < 0 ? 0.0 : p) : 0.0;
            }
            __syncthreads();
         }
      }
      else if (es.kind == PAML_AMD_EIGEN_CIJK) {
         uv_of = -1;
         double e[64];
         const int nR = es.nR;
         for (int idx = tid; idx < 64; idx += 256) sB[idx] = (idx >= 1 && idx < nR) ? expm1(t * es.Root[idx]) : 0.0;
         __syncthreads();
#pragma unroll
         for (int r = 0; r < ROWS; r++) {
            int i = rg * ROWS + r;
            double s = 0;
            if (i < n && j < n) {
               const double *c = es.Cijk + ((long)i * n + j) * nR;
               for (int k = 0; k < nR; k++) s += c[k] * sB[k];
               if (i == j) s += 1.0;
            }
            acc[r] = s;
         }
         (void)e;
         __syncthreads();
      }
      else if (es.kind == PAML_AMD_EIGEN_K80) {
         const double kappa = es.kappa;
         const double e1 = expm1(-4 * t / (kappa + 2));
         const bool jc = fabs(kappa - 1) < 1e-20;
         const double e2 = jc ? 0.0 : expm1(-2 * t * (kappa + 1) / (kappa + 2));
#pragma unroll
         for (int r = 0; r < ROWS; r++) {
            int i = rg * ROWS + r;
            double p = 0;
            if (i < 4 && j < 4) {
               if (jc) p = (i == j) ? 1. + 3 / 4.0 * e1 : -e1 / 4;
               else if (i == j) p = 1 + (e1 + 2 * e2) / 4;
               else if ((i ^ j) == 1) p = (e1 - 2 * e2) / 4;
               else p = -e1 / 4;
            }
            acc[r] = p;
         }
      }
      else if (es.kind == PAML_AMD_EIGEN_QMAT) {
         // UNREST: P = e^{Qt} by matexp(Qt, n, 7, 5) (tools.c:4879): B = Qt/32, e^B by seven Taylor terms, then five squarings.
         // n <= 8: one thread per entry, four n x n scratch matrices in LDS.
         double *T0 = sA, *T1 = sA + 64, *T2 = sA + 128, *Bm = sA + 192;
         const int i = tid / n, jj = tid % n;
         const bool on = tid < n * n;
         if (on) {
            const double v = es.U[tid] * t * (1.0 / 32);
            Bm[tid] = v; T1[tid] = v;
            T0[tid] = v + (i == jj ? 1.0 : 0.0);
         }
         __syncthreads();
         double factor = 1;
         double *Tp = T1, *Tn = T2;              // B^(k-1) and B^k
         for (int term = 2; term <= 7; term++) {
            double s = 0;
            if (on)
               for (int k2 = 0; k2 < n; k2++) s += Tp[i * n + k2] * Bm[k2 * n + jj];
            factor /= term;
            if (on) { Tn[tid] = s; T0[tid] += s * factor; }
            __syncthreads();
            double *sw = Tp; Tp = Tn; Tn = sw;
         }
         double *Sa = T0, *Sb = T1;
         for (int sq = 0; sq < 5; sq++) {
            double s = 0;
            if (on)
               for (int k2 = 0; k2 < n; k2++) s += Sa[i * n + k2] * Sa[k2 * n + jj];
            __syncthreads();
            if (on) Sb[tid] = s;
            __syncthreads();
            double *sw = Sa; Sa = Sb; Sb = sw;
         }
#pragma unroll
         for (int r = 0; r < ROWS; r++) {
            const int ii = rg * ROWS + r;
            acc[r] = (ii < n && j < n) ? Sa[ii * n + j] : 0.0;
         }
         __syncthreads();
      }
      else {   // JC69-like (aa Poisson): no Qfactor (treesub.c:7584-7585)
         const double pii = 1. / n + (1. - 1. / n) * exp(-n / (n - 1.) * t);
         const double pij = (1. - pii) / (n - 1.);
#pragma unroll
         for (int r = 0; r < ROWS; r++) {
            int i = rg * ROWS + r;
            acc[r] = (i < n && j < n) ? (i == j ? pii : pij) : 0.0;
         }
      }

      // finished P (zero padded to LD x LD) into LDS
#pragma unroll
      for (int r = 0; r < ROWS; r++) sA[(rg * ROWS + r) * LD + j] = acc[r];
      __syncthreads();

      const long slot = (long)pset * a.n_nodes + node;
      double *rm = a.rowmajor + slot * n * n;
      for (int idx = tid; idx < n * n; idx += 256) rm[idx] = sA[(idx / n) * LD + (idx % n)];

      if constexpr (LD == 32) if (a.layout == 2 && !leaf && a.pint) {
         // 20 states on the matrix cores (jit_generate_m20): the branch's 400 entries in the order the kernel's lanes consume them — rows
         // 0-15 as v_mfma_f64_16x16x4 A operands [kb][lane] = P[lane & 15][4 kb + (lane >> 4)], rows 16-19 as 4x4x4 operands
         // [kb][k][i] = P[16 + i][4 kb + k] — so that the pruning kernel fills its LDS with a plain copy and reads the branches LDS has no
         // room for straight from here (device_common.h: m20h_matvec2x)
         double *pf = a.pint + slot * 400;
         for (int idx = tid; idx < 400; idx += 256) {
            const int j = idx - 320;
            pf[idx] = idx < 320 ? sA[(idx & 15) * LD + 4 * (idx >> 6) + ((idx >> 4) & 3)] : sA[(16 + (j & 3)) * LD + 4 * (j >> 4) + ((j >> 2) & 3)];
         }
      }
      if constexpr (LD == 64) if ((a.layout == 1 || a.layout == 3) && !leaf) {
         // MFMA A-operand order: element ((kb2*4 + jb)*64 + lane)*2 + e  =  P[jb*16 + (lane&15)][4*(2*kb2+e) + (lane>>4)]
         // layout 3 (61 states, per-tree kernel without the row padding: device_common.h, JitRowTail): the fourth row block's slot of
         // every k-block pair holds the 4 x 4 x 4 operands of rows 48..59 and the pair's eight entries of row 60
         double *pf = a.pint + slot * 4096;
         for (int idx = tid; idx < 4096; idx += 256) {
            int e = idx & 1, lane = (idx >> 1) & 63, jb = (idx >> 7) & 3, kb2 = idx >> 9;
            double v = sA[(jb * 16 + (lane & 15)) * 64 + 4 * (2 * kb2 + e) + (lane >> 4)];
            if (a.layout == 3 && jb == 3) {
               const int li = idx & 127;
               if (li < 96) {
                  const int blk = li >> 4, w = li & 15;      // blk = e' * 3 + (m' - 12); w = k * 4 + i
                  v = sA[(4 * (12 + blk % 3) + (w & 3)) * 64 + 4 * (2 * kb2 + blk / 3) + (w >> 2)];
               }
               else if (li < 104) v = sA[60 * 64 + 4 * (2 * kb2 + (li & 1)) + ((li - 96) >> 1)];
               else v = 0.0;
            }
            pf[idx] = v;
         }
         // column 60 in the order a lane's accumulators want it, pcol[q][m] = P[4m + q][60]: with 61 states the last
         // k-block holds this one column, and the specialised kernel adds its rank-1 term on the vector pipe instead of
         // spending four MFMAs on it
         if (a.pcol && tid < 64) a.pcol[slot * 64 + tid] = sA[(4 * (tid & 15) + (tid >> 4)) * 64 + 60];
      }
      if (leaf) {
         const bool mf = a.layout == 1 || a.layout == 3;
         const int tipw = mf ? 64 : n;
         double *pt = a.ptip + slot * a.tip_words;
         for (int idx = tid; idx < a.n_codes * tipw; idx += 256) {
            int code = idx / tipw, w = idx % tipw, jj;
            if (mf) {
               // row (code, q) = 128 bytes = 8 pieces of two states; piece p is stored in slot p ^ ((row >> 1) & 7) so
               // that lanes gathering different rows from an LDS copy of this table spread over the banks
               const int q = w >> 4, slot = (w & 15) >> 1, row = code * 4 + q;
               const int m = ((slot ^ TIP_SWZ(row)) << 1) | (w & 1);
               jj = 4 * m + q;
            }
            else if (a.layout == 2) jj = 4 * (w % 5) + w / 5;      // 20 states on 4x4x4 MFMAs: [code][state & 3][state >> 2], a lane's five states contiguous
            else jj = w;
            double s = 0;
            if (jj < n) {
               const int nc = sNch[code];
               const unsigned char *map = sMap + code * n;
               for (int k = 0; k < nc; k++) s += sA[jj * LD + map[k]];
            }
            pt[idx] = s;
         }
      }
      __syncthreads();      // (the next node's matrices go over this one's)
   }
}


// ------------------------------------------------------------------------------------------------
// Kernel A on the matrix cores (round 4): the (U, V, Root) models of 21..64 states in the mfma64 layout — one workgroup per (node,
// parameter set, 16-ROW BLOCK of P), four per matrix.  pmat_kernel_t<64> takes 21 us for a matrix however few there are (13 taxa:
// 23 matrices on 256 CUs; profiles/r04_small_timeline.txt): U and V staged through LDS, a 61-step loop of LDS reads, the three
// output layouts gathered from LDS with 16-way bank conflicts, one workgroup's serial chain.  Here a wave owns a 16 x 16 block of
// the row block: its A operands are U's rows times expm1(t Root_k) and its B operands V's columns, both straight from global memory
// (L2) in the instruction's own lane order (the A operands, the same for the four waves, a quarter per wave and shared through LDS), sixteen v_mfma_f64_16x16x4 (k ascending, as PMatUVRoot accumulates, tools.c:525-537),
// `+ I`, the clamp, and the row block goes to LDS (row stride 65: conflict-free for the column-order outputs).  Every output
// layout is complete per row block — rowmajor rows, the prune kernels' A-operand order [k-block pair][ROW BLOCK][lane][2], the
// q-major entries of column 60, the rows jj = 4m + q of the tips' column tables — so the four workgroups of a matrix share nothing.
// ------------------------------------------------------------------------------------------------
typedef double pm_v2d __attribute__((ext_vector_type(2)));
// Large launches (launch_pmat: from 128 matrices on) send the output blocks out with streaming stores: they are read by the NEXT
// kernel, on every XCD, and through a write-back L2 they only start for memory at this kernel's end — 253 matrices (13 MB): the kernel
// 14.2 -> 11.6 us, the walk behind it 17.3 -> 18.8 (its first reads now miss), the evaluation 31.8 -> 30.4; at 23 matrices the walk loses
// what this kernel gains and more (20.7 -> 22.1), so small launches keep plain stores (profiles/r05_pmat_phases.txt).
// (two instantiations: behind a run-time flag the compiler merges the two stores into the plain one)
#define PMAT_ST2(P, X, Y) do { if (NT) __builtin_nontemporal_store((pm_v2d){X, Y}, (pm_v2d *)(P)); else *(pm_v2d *)(P) = (pm_v2d){X, Y}; } while (0)
template <bool NT>
__global__ __launch_bounds__(256) void pmat_mfma_kernel(PmatArgs a, InlineVec iv)
{
   __shared__ double sE[64];
   __shared__ double sP[16 * 65];
   __shared__ double sA[16 * 64];      // the row block's A operands [k-block][lane]: every wave needs all of them, each loads a quarter
   __shared__ unsigned char sMap[256 * 64];
   __shared__ int sNch[256];
   const int node = blockIdx.x, pset = blockIdx.y, rb = blockIdx.z;
   if (node == a.root) return;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = a.n;
   const int KB = a.K * a.B;
   const int gene = pset / KB, bat = (pset % KB) / a.K, iclass = pset % a.K;
   EigenDev es;
   double t;
   bool leaf;
   if (a.res) {      // resolved on the host (single evaluations; branch lengths and gene rates ride in the kernel arguments)
      const PmatRes r = a.res[(long)pset * a.n_nodes + node];
      es.U = r.U; es.V = r.V; es.Root = r.Root;
      t = ((iv.v[node] * r.rate) * iv.v[iv.n_branch + gene]) * r.qfactor;
      leaf = r.leaf != 0;
   }
   else {
      const int lab = a.label[node];
      es = a.eigen[a.eigen_of[bat * a.eigen_of_bs + (gene * a.K + iclass) * a.n_labels + lab]];
      t = pmat_time(a, iv, bat, node, gene, iclass) * a.qfactor[bat * a.qfactor_bs + iclass * a.n_labels + lab];
      leaf = a.is_leaf[node] != 0;
   }
   // this lane's operands of the sixteen k-blocks: A = U[16 rb + (lane & 15)][4 kb + (lane >> 4)], B = V[4 kb + (lane >> 4)][16 wave + (lane & 15)]
   const int ai = rb * 16 + (lane & 15), bj = wave * 16 + (lane & 15), kq = lane >> 4;
   // (the four waves' A operands are the same sixteen values per lane: wave w fetches k-blocks 4 w .. 4 w + 3 and they are shared through
   //  LDS — 40 KB instead of 64 KB of L2 reads per workgroup, which is what a launch of a thousand workgroups (M8: 253 matrices) waits for)
   double ua4[4], vb[16];
#pragma unroll
   for (int kb = 0; kb < 16; kb++) {
      const int k = 4 * kb + kq;
      vb[kb] = (bj < n && k < n) ? es.V[k * n + bj] : 0.0;
   }
#pragma unroll
   for (int i = 0; i < 4; i++) {
      const int k = 4 * (4 * wave + i) + kq;
      ua4[i] = (ai < n && k < n) ? es.U[ai * n + k] : 0.0;
   }
   if (tid < 64) sE[tid] = (tid < n && !(t < 1e-100)) ? expm1(t * es.Root[tid]) : 0.0;      // (t < 1e-100: P = I, tools.c:521)
   // the ambiguity map (tools.c:20 nChara / CharaMap) for the column tables below — only where there ARE ambiguity codes: the first
   // a.plain_codes codes are single states equal to the code (all of them with cleandata), and staging 3.7 KB byte by byte in front of
   // the matrix product was 1.8 of the kernel's 6.8 us at 23 matrices, 2.4 of 16.8 at 253 (profiles/r05_pmat_phases.txt)
   const int plain = a.plain_codes;
   if (leaf && a.n_codes > plain) {
      for (int idx = tid; idx < a.n_codes * n; idx += 256) sMap[idx] = a.chara_map[idx];
      for (int idx = tid; idx < a.n_codes; idx += 256) sNch[idx] = a.n_chara[idx];
   }
#pragma unroll
   for (int i = 0; i < 4; i++) sA[(4 * wave + i) * 64 + lane] = ua4[i];
   __syncthreads();
   typedef double pm_v4d __attribute__((ext_vector_type(4)));
   pm_v4d acc = {0, 0, 0, 0};
#pragma unroll
   for (int kb = 0; kb < 16; kb++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[kb * 64 + lane] * sE[4 * kb + kq], vb[kb], acc, 0, 0, 0);
   // accumulator element r of this lane = P[16 rb + 4 r + (lane >> 4)][16 wave + (lane & 15)]
#pragma unroll
   for (int r = 0; r < 4; r++) {
      const int il = 4 * r + kq, i = rb * 16 + il;
      double p = acc[r] + (i == bj ? 1.0 : 0.0);
      p = (i < n && bj < n) ? (p < 0 ? 0.0 : p) : 0.0;
      sP[il * 65 + bj] = p;
   }
   __syncthreads();

   const long slot = (long)pset * a.n_nodes + node;
   if (a.rowmajor) {      // (null: nobody reads it — the pruning kernels take the operand-order copies below, and paml_amd_get_pmat rebuilds P from those)
      double *rm = a.rowmajor + slot * n * n;
      for (int idx = tid; idx < 16 * n; idx += 256) {
         const int il = idx / n, j = idx % n, i = rb * 16 + il;
         if (i < n) rm[i * n + j] = sP[il * 65 + j];
      }
   }
   if (!leaf) {
      // MFMA A-operand order: element ((kb2*4 + jb)*64 + lane)*2 + e  =  P[jb*16 + (lane&15)][4*(2*kb2+e) + (lane>>4)]; jb = rb here
      double *pf = a.pint + slot * 4096;
      for (int idx = tid; idx < 512; idx += 256) {      // (a lane's pair e = 0, 1 is one 16-byte store)
         const int ln = idx & 63, kb2 = idx >> 6;
         const double *sp = sP + (ln & 15) * 65 + 8 * kb2 + (ln >> 4);
         PMAT_ST2(pf + ((kb2 * 4 + rb) * 64 + ln) * 2, sp[0], sp[4]);
      }
      // column 60 as pcol[q][m] = P[4m + q][60] (the per-tree kernel's rank-1 term): the rows of this block
      if (a.pcol && tid < 64) {
         const int row = 4 * (tid & 15) + (tid >> 4);
         if ((row >> 4) == rb) a.pcol[slot * 64 + tid] = sP[(row & 15) * 65 + 60];
      }
   }
   else {
      // column sums over each character code's state set (codeml.c:3555-3567), table [code][q][slot] with the XOR swizzle of the
      // pieces (pmat_kernel_t): the entries whose row jj = 4m + q lies in this block
      double *pt = a.ptip + slot * a.tip_words;
      for (int idx = tid; idx < a.n_codes * 8; idx += 256) {      // (rows jj and jj + 4 — m and m + 1, same q — are neighbours: one 16-byte store)
         const int code = idx >> 3, q = idx & 3, mh = (idx >> 2) & 1, il = q + 8 * mh, jj = rb * 16 + il, m = jj >> 2;
         const int row = code * 4 + q, w = q * 16 + ((((m >> 1) ^ TIP_SWZ(row)) & 7) << 1);
         double s2[2] = {0, 0};
#pragma unroll
         for (int h = 0; h < 2; h++) {
            const int ilh = il + 4 * h;
            if (jj + 4 * h >= n) continue;
            if (code < plain) s2[h] = sP[ilh * 65 + code];
            else {
               const int nc = sNch[code];
               const unsigned char *map = sMap + code * n;
               for (int k = 0; k < nc; k++) s2[h] += sP[ilh * 65 + map[k]];
            }
         }
         PMAT_ST2(pt + code * 64 + w, s2[0], s2[1]);
      }
   }
}

}  // namespace paml_amd
