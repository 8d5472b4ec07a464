// kernels.h — hand-written HIP kernels for gfx950 (MI355X, CDNA4): batched P(t) construction,
// fused Felsenstein pruning (FP64 MFMA path for 21..64 states, per-pattern VALU path for 4/5/20
// states) and the deterministic root/mixture/log reduction.  Wave = 64 lanes throughout.
#pragma once
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "program.h"

namespace paml_amd {


struct EigenDev {
   int kind, nR;
   double kappa;
   const double *U, *V, *Root, *Cijk;
};

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
struct PmatArgs {
   int n, n_nodes, root, K, n_genes, n_labels, n_codes, layout;   // layout 0: VALU (row-major), 1: mfma64, 2: as 0 with the tip rows in m20 order
   const int *label;             // [n_nodes]
   const unsigned char *is_leaf; // [n_nodes]
   const double *branch;         // [n_nodes]
   const double *rate;           // [K]
   const double *gene_rate;      // [n_genes]
   const int *eigen_of;          // [n_genes][K][n_labels]
   const double *qfactor;        // [K][n_labels]
   const EigenDev *eigen;
   const int *n_chara;           // [n_codes]
   const unsigned char *chara_map; // [n_codes][n]
   double *rowmajor;             // [pset][n_nodes][n*n]
   double *pint;                 // layout 1: [pset][n_nodes][4096]
   double *ptip;                 // [pset][n_nodes][tip_words]   rows of n (VALU) or 64 (mfma64) doubles per code
   long tip_words;
   double *pcol;                 // layout 1: [pset][n_nodes][64], column 60 per (q, m) (null: not wanted)
   // batched evaluations (paml_amd_eval_batch): B parameter sets in one launch, laid out as K*B classes; element b reads
   // branch + b*branch_bs etc. (a stride of 0 = shared with the other elements)
   int B;
   long branch_bs, gene_rate_bs, eigen_of_bs, qfactor_bs, rate_bs;
   int rate_gs;                   // class rates per gene (Malpha: a gamma shape per gene): rate[bat][gene][class], else 0
};

// Branch lengths and gene rates handed over INSIDE the kernel arguments (single evaluations of trees with up to ~440 nodes):
// the launch itself carries them, so an evaluation needs no host-to-device copy and no staging buffer to keep alive.
#define PMAT_INLINE_MAX 440
struct InlineVec {
   int n_branch, n_rate;          // 0, 0: read PmatArgs::branch / gene_rate instead
   double v[PMAT_INLINE_MAX];     // branch[n_branch], then gene_rate[n_rate]
};

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

__global__ __launch_bounds__(256) void pmat_kernel(PmatArgs a, InlineVec iv)
{
   extern __shared__ __attribute__((aligned(16))) double smem[];
   double *sA = smem;            // [64][64]  U*expm1 -> later the finished P (padded with zeros)
   double *sB = smem + 4096;     // [64][64]  V
   const int node = blockIdx.x, pset = blockIdx.y;
   if (node == a.root) return;
   const int tid = threadIdx.x, n = a.n;
   const int KB = a.K * a.B;
   const int gene = pset / KB, bat = (pset % KB) / a.K, iclass = pset % a.K;
   const int lab = a.label[node];
   const EigenDev es = a.eigen[a.eigen_of[bat * a.eigen_of_bs + (gene * a.K + iclass) * a.n_labels + lab]];
   double t = pmat_time(a, iv, bat, node, gene, iclass);

   // tip branches: the ambiguity map (tools.c:20 nChara / CharaMap) comes to LDS now, so that the column-table loop at
   // the end does not chase two dependent global loads per entry
   __shared__ unsigned char sMap[256 * 64];
   __shared__ int sNch[256];
   const bool leaf = a.is_leaf[node] != 0;
   if (leaf) {
      for (int idx = tid; idx < a.n_codes * n; idx += 256) sMap[idx] = a.chara_map[idx];
      for (int idx = tid; idx < a.n_codes; idx += 256) sNch[idx] = a.n_chara[idx];
   }

   const int j = tid & 63, rg = tid >> 6;   // this thread: column j, rows rg*16 .. rg*16+15
   double acc[16];
#pragma unroll
   for (int r = 0; r < 16; r++) acc[r] = 0;

   if (es.kind == PAML_AMD_EIGEN_UVROOT) {
      t *= a.qfactor[bat * a.qfactor_bs + iclass * a.n_labels + lab];
      if (t < 1e-100) {
#pragma unroll
         for (int r = 0; r < 16; r++) acc[r] = (rg * 16 + r == j) ? 1.0 : 0.0;
      }
      else {
         for (int idx = tid; idx < 4096; idx += 256) {
            int i = idx >> 6, k = idx & 63;
            double ue = 0, v = 0;
            if (i < n && k < n) {
               ue = es.U[i * n + k] * expm1(t * es.Root[k]);
               v = es.V[i * n + k];        // here (i,k) index V as [k'][j'] = [i][k]
            }
            sA[k * 64 + i] = ue;           // transposed: the four rows of a register tile are contiguous for every k
            sB[idx] = v;
         }
         __syncthreads();
         {
            // 4 x 4 register tile per thread: four 16-byte LDS reads feed sixteen FMAs; every element still accumulates
            // k ascending (PMatUVRoot's order, tools.c:525-537)
            const int ti = tid >> 4, tj = tid & 15;
            double c[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
               for (int cc = 0; cc < 4; cc++) c[r][cc] = 0;
            for (int k = 0; k < n; k++) {
               const double2 a0 = *(const double2 *)&sA[k * 64 + 4 * ti], a1 = *(const double2 *)&sA[k * 64 + 4 * ti + 2];
               const double2 b0 = *(const double2 *)&sB[k * 64 + 4 * tj], b1 = *(const double2 *)&sB[k * 64 + 4 * tj + 2];
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
               for (int cc = 0; cc < 4; cc++) sA[(4 * ti + r) * 64 + 4 * tj + cc] = c[r][cc];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = sA[(rg * 16 + r) * 64 + j];
         }
#pragma unroll
         for (int r = 0; r < 16; r++) {
            int i = rg * 16 + r;
            double p = acc[r] + (i == j ? 1.0 : 0.0);
            acc[r] = (i < n && j < n) ? (p < 0 ? 0.0 : p) : 0.0;
         }
         __syncthreads();
      }
   }
   else if (es.kind == PAML_AMD_EIGEN_CIJK) {
      double e[64];
      const int nR = es.nR;
      for (int idx = tid; idx < 64; idx += 256) sB[idx] = (idx >= 1 && idx < nR) ? expm1(t * es.Root[idx]) : 0.0;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; r++) {
         int i = rg * 16 + r;
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
      for (int r = 0; r < 16; r++) {
         int i = rg * 16 + r;
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
      for (int r = 0; r < 16; r++) {
         const int ii = rg * 16 + r;
         acc[r] = (ii < n && j < n) ? Sa[ii * n + j] : 0.0;
      }
      __syncthreads();
   }
   else {   // JC69-like (aa Poisson): no Qfactor (treesub.c:7584-7585)
      const double pii = 1. / n + (1. - 1. / n) * exp(-n / (n - 1.) * t);
      const double pij = (1. - pii) / (n - 1.);
#pragma unroll
      for (int r = 0; r < 16; r++) {
         int i = rg * 16 + r;
         acc[r] = (i < n && j < n) ? (i == j ? pii : pij) : 0.0;
      }
   }

   // finished P (zero padded to 64x64) into LDS
#pragma unroll
   for (int r = 0; r < 16; r++) sA[(rg * 16 + r) * 64 + j] = acc[r];
   __syncthreads();

   const long slot = (long)pset * a.n_nodes + node;
   double *rm = a.rowmajor + slot * n * n;
   for (int idx = tid; idx < n * n; idx += 256) rm[idx] = sA[(idx / n) * 64 + (idx % n)];

   if (a.layout == 1 && !leaf) {
      // MFMA A-operand order: element ((kb2*4 + jb)*64 + lane)*2 + e  =  P[jb*16 + (lane&15)][4*(2*kb2+e) + (lane>>4)]
      double *pf = a.pint + slot * 4096;
      for (int idx = tid; idx < 4096; idx += 256) {
         int e = idx & 1, lane = (idx >> 1) & 63, jb = (idx >> 7) & 3, kb2 = idx >> 9;
         pf[idx] = sA[(jb * 16 + (lane & 15)) * 64 + 4 * (2 * kb2 + e) + (lane >> 4)];
      }
      // column 60 in the order a lane's accumulators want it, pcol[q][m] = P[4m + q][60]: with 61 states the last
      // k-block holds this one column, and the specialised kernel adds its rank-1 term on the vector pipe instead of
      // spending four MFMAs on it
      if (a.pcol && tid < 64) a.pcol[slot * 64 + tid] = sA[(4 * (tid & 15) + (tid >> 4)) * 64 + 60];
   }
   if (leaf) {
      const int tipw = a.layout == 1 ? 64 : n;
      double *pt = a.ptip + slot * a.tip_words;
      for (int idx = tid; idx < a.n_codes * tipw; idx += 256) {
         int code = idx / tipw, w = idx % tipw, jj;
         if (a.layout == 1) {
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
            for (int k = 0; k < nc; k++) s += sA[jj * 64 + map[k]];
         }
         pt[idx] = s;
      }
   }
}

// ------------------------------------------------------------------------------------------------
// Pruning kernels: arguments shared by the MFMA and VALU variants.
// ------------------------------------------------------------------------------------------------
// ---- mfma64: 21..64 states, FP64 MFMA -----------------------------------------------------------
// One wave owns 16 patterns for the whole tree.  A partial is 16 doubles per lane: lane l holds, for
// pattern (l & 15), the states 4m + (l >> 4), m = 0..15.  That is simultaneously
//   * the B operand of v_mfma_f64_16x16x4_f64 for k-block kb = m  (B[k = l>>4][n = l&15]), and
//   * the D layout of the instruction for row block jb = m>>2, register m&3  (row = (l>>4) + 4 reg),
// so cur' = P . cur chains from node to node entirely in registers: no transposes, no LDS traffic for
// partials.  The A operand (P) is staged once per workgroup per branch into LDS in exactly the order
// lanes consume it (pmat_kernel's `frag` layout), double-buffered so the next branch's P streams in
// under the current MFMAs.  Tip branches are gathers from L2-resident column tables.
// Shared op bodies of the two mfma64 kernels (textual, so every register array keeps static indices).
#define MFMA_EPI_INTO(DST)                                                                                       \
   do {                                                                                                         \
      if (pop < 0) {                                                                                            \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = acc[m >> 2][m & 3];                            \
      }                                                                                                         \
      else if (pop == 0) {                                                                                      \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = s0[m] * acc[m >> 2][m & 3];                    \
      }                                                                                                         \
      else if (pop == 1) {                                                                                      \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = s1[m] * acc[m >> 2][m & 3];                    \
      }                                                                                                         \
      else {                                                                                                    \
         const double *sp2 = a.stack_scratch +                                                                  \
                             (((long)blockIdx.x * a.stack_overflow_slots + (pop - MFMA_RS)) * WAVES + wave) * 1024; \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = sp2[m * 64 + lane] * acc[m >> 2][m & 3];       \
      }                                                                                                         \
   } while (0)

#define MFMA_EPILOGUE()                                                                                          \
   do {                                                                                                         \
      const int pop = mm_pop_slot(op), push = mm_push_slot(op);                                                 \
      if (push < 0) MFMA_EPI_INTO(cur);                                                                         \
      else if (push == 0) MFMA_EPI_INTO(s0);                                                                    \
      else if (push == 1) MFMA_EPI_INTO(s1);                                                                    \
      else {                                                                                                    \
         double tmpv[16];                                                                                       \
         MFMA_EPI_INTO(tmpv);                                                                                   \
         double *sp3 = a.stack_scratch +                                                                        \
                       (((long)blockIdx.x * a.stack_overflow_slots + (push - MFMA_RS)) * WAVES + wave) * 1024;  \
         _Pragma("unroll") for (int m = 0; m < 16; m++) sp3[m * 64 + lane] = tmpv[m];                           \
      }                                                                                                         \
   } while (0)

#define MFMA_CORE_CASES()                                                                                        \
   case OP_INIT_ONES: {                                                                                         \
      _Pragma("unroll") for (int m = 0; m < 16; m++) cur[m] = (4 * m + q < n) ? 1.0 : 0.0;                      \
   } break;                                                                                                     \
   case OP_INIT_TIP: {                                                                                          \
      const int code = TIP_CODE(op.a);                                                                          \
      _Pragma("unroll") for (int m = 0; m < 16; m++) cur[m] = (a.cleandata && 4 * m + q == code) ? 1.0 : 0.0;   \
   } break;

#define MFMA_EXT_CASES()                                                                                         \
   case OP_PUSH: {                                                                                              \
      if (op.b == 0) { _Pragma("unroll") for (int m = 0; m < 16; m++) s0[m] = cur[m]; }                         \
      else if (op.b == 1) { _Pragma("unroll") for (int m = 0; m < 16; m++) s1[m] = cur[m]; }                    \
      else {                                                                                                    \
         double *sp = a.stack_scratch +                                                                         \
                      (((long)blockIdx.x * a.stack_overflow_slots + (op.b - MFMA_RS)) * WAVES + wave) * 1024;   \
         _Pragma("unroll") for (int m = 0; m < 16; m++) sp[m * 64 + lane] = cur[m];                             \
      }                                                                                                         \
   } break;                                                                                                     \
   case OP_SCALE: {                                                                                             \
      double mx = 0;                                                                                            \
      _Pragma("unroll") for (int m = 0; m < 16; m++) mx = cur[m] > mx ? cur[m] : mx;                            \
      double o = __shfl_xor(mx, 16);                                                                            \
      mx = o > mx ? o : mx;                                                                                     \
      o = __shfl_xor(mx, 32);                                                                                   \
      mx = o > mx ? o : mx;                                                                                     \
      double fac;                                                                                               \
      if (mx < 1e-300) {                                                                                        \
         _Pragma("unroll") for (int m = 0; m < 16; m++) cur[m] = (4 * m + q < n) ? 1.0 : 0.0;                   \
         fac = -800;                                                                                            \
      }                                                                                                         \
      else {                                                                                                    \
         _Pragma("unroll") for (int m = 0; m < 16; m++) cur[m] /= mx;                                           \
         fac = log(mx);                                                                                         \
      }                                                                                                         \
      lnscale += fac;                                                                                           \
      if (a.keep && q == 0 && valid) a.scalef[((long)iclass * a.n_scale + op.b) * a.n_patt + h] = fac;          \
   } break;                                                                                                     \
   case OP_STORE: {  /* native layout [class][node][16-pattern group][m][lane] */                               \
      double *dst = a.partials + (((long)iclass * a.n_int + (op.a - a.n_tips)) * ((long)a.n_tiles * WAVES) +    \
                                  ((long)tile * WAVES + wave)) * 1024;                                          \
      _Pragma("unroll") for (int m = 0; m < 16; m++) dst[m * 64 + lane] = cur[m];                               \
   } break;                                                                                                     \
   case OP_LOAD: {                                                                                              \
      const double *src = a.partials + (((long)iclass * a.n_int + (op.a - a.n_tips)) * ((long)a.n_tiles * WAVES) + \
                                        ((long)tile * WAVES + wave)) * 1024;                                    \
      _Pragma("unroll") for (int m = 0; m < 16; m++) cur[m] = src[m * 64 + lane];                               \
   } break;

#define MFMA_ROOT_CASE()                                                                                         \
   case OP_ROOT: {                                                                                              \
      const double *pq = a.pi + (long)(a.n_pi > 1 ? gene : 0) * 64 + q * 16;                                    \
      double f = 0;                                                                                             \
      _Pragma("unroll") for (int m = 0; m < 16; m++) f = fma(pq[m], cur[m], f);                                 \
      f += __shfl_xor(f, 16);                                                                                   \
      f += __shfl_xor(f, 32);                                                                                   \
      if (a.keep && a.n_scale) { /* stored factors summed in slot order (treesub.c:7746-7747) */                \
         lnscale = 0;                                                                                           \
         if (valid)                                                                                             \
            for (int k = 0; k < a.n_scale; k++) lnscale += a.scalef[((long)iclass * a.n_scale + k) * a.n_patt + h]; \
      }                                                                                                         \
      if (q == 0 && valid) {                                                                                    \
         double out = 0;                                                                                        \
         if (a.weights[h] > 0) out = root_value(a, f, lnscale);                                                 \
         a.fhK[(long)iclass * a.n_patt + h] = out;                                                              \
      }                                                                                                         \
   } break;

// register-stack-only epilogue (programs with max_stack <= MFMA_RS)
#define MFMA_EPI_REG(DST)                                                                                        \
   do {                                                                                                         \
      if (pop < 0) {                                                                                            \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = acc[m >> 2][m & 3];                            \
      }                                                                                                         \
      else if (pop == 0) {                                                                                      \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = s0[m] * acc[m >> 2][m & 3];                    \
      }                                                                                                         \
      else {                                                                                                    \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = s1[m] * acc[m >> 2][m & 3];                    \
      }                                                                                                         \
   } while (0)
#define MFMA_EPILOGUE_REG()                                                                                      \
   do {                                                                                                         \
      const int pop = mm_pop_slot(op), push = mm_push_slot(op);                                                 \
      if (push < 0) MFMA_EPI_REG(cur);                                                                          \
      else if (push == 0) MFMA_EPI_REG(s0);                                                                     \
      else MFMA_EPI_REG(s1);                                                                                    \
   } while (0)

#ifdef PROF_OPS
#define PROF_STAMP(slot) \
   if (a.prof && tid == a.prof_tid) a.prof[(long)blockIdx.x * a.prof_stride + (slot)] = __builtin_amdgcn_s_memtime()
// sub-stamps inside an op: plane 1 / 2 of the dump (same [block][op] indexing)
#define PROF_SUB(plane, ip) \
   if (a.prof && tid == a.prof_tid) a.prof[((long)(plane)*gridDim.x + blockIdx.x) * a.prof_stride + 1 + (ip)] = __builtin_amdgcn_s_memtime()
#else
#define PROF_STAMP(slot)
#define PROF_SUB(plane, ip)
#endif

// ---- mfma64 "gather": tip columns gathered straight from the L2-resident tables into registers.
// Used for trees with more than MFMA_ZT tips; 4 waves (64 patterns) per workgroup, 2 workgroups per CU.
__device__ __forceinline__ void tip_gather(const double *Ptip, long tipstride, int tip, int code, int q, double2 (&v)[8])
{
   const int row = code * 4 + q, swz = TIP_SWZ(row);
   const double2 *pt = (const double2 *)(Ptip + (long)tip * tipstride + row * 16);
#pragma unroll
   for (int i = 0; i < 8; i++) v[i] = pt[i ^ swz];     // piece i lives in slot i ^ swz (see pmat_kernel)
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void prune_mfma64_gather(PruneArgs a)
{
   __shared__ __attribute__((aligned(16))) double sP[2][4096];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, hl = lane & 15;
   const int tile = blockIdx.x % a.n_tiles, iclass = blockIdx.x / a.n_tiles;
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y;
   const int hend = as_const(a.gene_off)[gene + 1];
   const int h = h0 + wave * 16 + hl;
   const bool valid = h < hend;
   const int hc = valid ? h : hend - 1;
   const long pset = (long)gene * a.K + iclass;
   const double *Pint = a.pint + pset * a.n_nodes * 4096;
   const long tipstride = a.tip_words;
   const double *Ptip = a.ptip + pset * a.n_nodes * tipstride;
   const int n = a.n;

   PROF_STAMP(a.prof_stride - 1);
   if (a.first_matmul >= 0) stage_p<WAVES>(Pint + (long)a.first_matmul * 4096, sP[0], wave, lane);

   double cur[16], s0[16], s1[16];   // every program writes cur/s0/s1 (INIT/SET/PUSH) before reading them
   double lnscale = 0;
   int buf = 0;
#define TIP_CODE(tip) ((int)a.z[(long)(tip)*a.z_stride + hc])
   PROF_STAMP(0);
   const int lane0 = lane;
   for (int ip = 0;; ip++) {
      const Op op = fetch_op(a.ops, ip);
      PROF_STAMP(1 + ip);
      if (op.code == OP_END) break;
      int lane = lane0;             // opaque per-iteration copy: keeps LICM from hoisting (and spilling) lane math
      asm volatile("" : "+v"(lane));
      const int q = lane >> 4;
      switch (op.code) {
         MFMA_CORE_CASES()
         MFMA_EXT_CASES()
         MFMA_ROOT_CASE()
      case OP_EXPORT: {
         if (valid) {
            double *dst = a.export_buf + ((long)iclass * a.n_patt + h) * n;
#pragma unroll
            for (int m = 0; m < 16; m++)
               if (4 * m + q < n) dst[4 * m + q] = cur[m];
            if (a.export_scale && q == 0) a.export_scale[(long)iclass * a.n_patt + h] = lnscale;
         }
      } break;
      case OP_MUL_TIP:
      case OP_SET_TIP: {
         double2 v[8];
         tip_gather(Ptip, tipstride, op.a, TIP_CODE(op.a), q, v);
         if (op.code == OP_SET_TIP) {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] = v[i].x; cur[2 * i + 1] = v[i].y; }
         }
         else {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] *= v[i].x; cur[2 * i + 1] *= v[i].y; }
         }
      } break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2: {
         const int c1 = TIP_CODE(op.a), c2 = TIP_CODE(op.b);
         double2 v[8], w[8];
         tip_gather(Ptip, tipstride, op.a, c1, q, v);
         tip_gather(Ptip, tipstride, op.b, c2, q, w);
         if (op.code == OP_SET_TIP2) {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] = v[i].x * w[i].x; cur[2 * i + 1] = v[i].y * w[i].y; }
         }
         else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
               cur[2 * i] = (cur[2 * i] * v[i].x) * w[i].x;
               cur[2 * i + 1] = (cur[2 * i + 1] * v[i].y) * w[i].y;
            }
         }
      } break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         __syncthreads();   // this branch's P has landed in sP[buf]; every wave is done reading sP[buf^1]
         if (op.c >= 0) stage_p<WAVES>(Pint + (long)op.c * 4096, sP[buf ^ 1], wave, lane);
         v4d acc[4];
         mfma_matvec(sP[buf], lane, cur, acc);
         MFMA_EPILOGUE();
         buf ^= 1;
      } break;
      default: break;
      }
   }
#undef TIP_CODE
}

// ---- mfma64 "stream": the production kernel (<= MFMA_ZT tips, <= 64 character codes, register stack).
// Every operand the tree walk consumes — the P of an internal branch in MFMA order, or the whole column
// table of a tip branch — is one 32 KB block, and the program fixes the order in which blocks are used.
// 8 waves (128 patterns) per workgroup share a ring of four 32 KB LDS buffers that a linear LDS-DMA
// stream keeps filled three blocks ahead of use (4 x buffer_load_dwordx4 ... lds per wave per block),
// so neither P nor tip data is ever waited for at L2 latency, no VGPRs hold data in flight, and every
// DMA instruction is a fully coalesced 1 KB line burst.  Tip factors are then LDS gathers (rows are
// XOR-swizzled by pmat_kernel so random rows spread over the banks); one s_barrier per step.
__global__ __launch_bounds__(512, 2) void prune_mfma64_stream(PruneArgs a)
{
   constexpr int WAVES = 8, TP = 128;
   extern __shared__ __attribute__((aligned(16))) unsigned char smem_stream[];
   double *ring = (double *)smem_stream;                        // [4][4096]
   unsigned char *sZ = (unsigned char *)(ring + 4 * 4096);      // [n_tips][128]
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int hl = lane & 15;
   const int tile = blockIdx.x % a.n_tiles, iclass = blockIdx.x / a.n_tiles;
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y;
   const int hend = as_const(a.gene_off)[gene + 1];
   const int hw = wave * 16 + hl;          // pattern within the tile
   const int h = h0 + hw;
   const bool valid = h < hend;
   const long pset = (long)gene * a.K + iclass;
   const double *Pint = a.pint + pset * a.n_nodes * 4096;
   const long tipstride = 4096;            // one 32 KB block per tip (n_codes <= 64)
   const double *Ptip = a.ptip + pset * a.n_nodes * tipstride;
   const int n = a.n;
   const StreamBlk *stream = (const StreamBlk *)a.stream;
   const int nblk = a.n_stream;

   PROF_STAMP(a.prof_stride - 1);
   int issued = 0, consumed = 0;
#define STREAM_ISSUE()                                                                                           \
   do {                                                                                                         \
      const long long sb = ((const CONST_AS long long *)(unsigned long long)stream)[issued];                    \
      const int is_tip = (int)(sb & 0xffffffff), node = (int)(sb >> 32);                                        \
      const double *src = is_tip ? Ptip + (long)node * tipstride : Pint + (long)node * 4096;                    \
      stage_p<WAVES>(src, ring + (issued & 3) * 4096, wave, lane);                                              \
      issued++;                                                                                                 \
   } while (0)
   for (int i = 0; i < 3; i++)
      if (issued < nblk) STREAM_ISSUE();
   {
      const int nz = a.n_tips * TP;
      for (int idx = tid; idx < nz; idx += WAVES * 64) {
         const int tip = idx / TP, hh = idx % TP;
         const int hx = h0 + hh < hend ? h0 + hh : hend - 1;
         sZ[idx] = a.z[(long)tip * a.z_stride + hx];
      }
   }
   __syncthreads();   // publish sZ (INIT_TIP may read it before the first stream step)

   double cur[16], s0[16], s1[16];   // every program writes cur/s0/s1 (INIT/SET/PUSH) before reading them
   double lnscale = 0;
#define TIP_CODE(tip) ((int)sZ[(tip)*TP + hw])
   // consume the next c blocks of the stream: they have landed for every wave after this returns, and the
   // buffers used by the previous step are refilled with the blocks 3..4 ahead
#ifdef ABL_NO_BARRIER
#define STREAM_BARRIER()
#else
#define STREAM_BARRIER() __syncthreads()
#endif
#define STREAM_STEP(c)                                                                                           \
   do {                                                                                                         \
      wait_blocks_in_flight(issued - (consumed + (c)));                                                         \
      STREAM_BARRIER();                                                                                         \
      while (issued < consumed + 4 && issued < nblk) STREAM_ISSUE();                                            \
   } while (0)

   PROF_STAMP(0);
   const int lane0 = lane;
   for (int ip = 0;; ip++) {
      const Op op = fetch_op(a.ops, ip);
      PROF_STAMP(1 + ip);
      if (op.code == OP_END) break;
      // re-derive every per-lane address from an opaque copy of the lane id inside the loop: cheap VALU,
      // and nothing loop-invariant is left for LICM to hoist into (spilled) VGPRs
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      const int q = lane >> 4;
      switch (op.code) {
         MFMA_CORE_CASES()
         MFMA_ROOT_CASE()
      case OP_MUL_TIP:
      case OP_SET_TIP: {
         double2 v[8];
         STREAM_STEP(1);
         tip_lds(ring + (consumed & 3) * 4096, TIP_CODE(op.a), q, lane, v);
         consumed += 1;
         if (op.code == OP_SET_TIP) {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] = v[i].x; cur[2 * i + 1] = v[i].y; }
         }
         else {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] *= v[i].x; cur[2 * i + 1] *= v[i].y; }
         }
      } break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2: {
         double2 v[8], w[8];
         STREAM_STEP(2);
         tip_lds(ring + (consumed & 3) * 4096, TIP_CODE(op.a), q, lane, v);
         tip_lds(ring + ((consumed + 1) & 3) * 4096, TIP_CODE(op.b), q, lane, w);
         consumed += 2;
         if (op.code == OP_SET_TIP2) {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] = v[i].x * w[i].x; cur[2 * i + 1] = v[i].y * w[i].y; }
         }
         else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
               cur[2 * i] = (cur[2 * i] * v[i].x) * w[i].x;
               cur[2 * i + 1] = (cur[2 * i + 1] * v[i].y) * w[i].y;
            }
         }
      } break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         STREAM_STEP(1);
         PROF_SUB(1, ip);
         v4d acc[4];
         mfma_matvec(ring + (consumed & 3) * 4096, lane, cur, acc);
         consumed += 1;
         PROF_SUB(2, ip);
         MFMA_EPILOGUE_REG();
      } break;
      default: break;
      }
   }
#undef TIP_CODE
#undef STREAM_STEP
#undef STREAM_ISSUE
}

// ---- valu<N>: 4 / 5 / 20 states, one pattern per lane ------------------------------------------
// The partial lives in N registers; P(t) entries are wave-uniform, so the compiler fetches them with
// scalar loads (s_load) and feeds v_fma_f64 from SGPRs: no LDS, no barriers.  The whole tree is walked
// per lane, so only tips (1 B) and the result (8 B) touch HBM unless keep-partials is on.
// REGSTK: the partial stack is addressed through unrolled wave-uniform compares, so it stays in registers; the plain
// form indexes stk[op.b] dynamically, which the compiler can only do through scratch memory (measured on the 20-state
// kernel: 680 MB of scratch writes per launch at 1e5 patterns).  The engine picks the shallowest instantiation that
// fits the tree's stack depth.
template <int N, int MAXD, bool REGSTK = false>
__global__ __launch_bounds__(256) void prune_valu(PruneArgs a)
{
   const int tid = threadIdx.x;
   const int tile = blockIdx.x % a.n_tiles, iclass = blockIdx.x / a.n_tiles;
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y;
   const int hend = as_const(a.gene_off)[gene + 1];
   const int h = h0 + tid;
   const bool valid = h < hend;
   const int hc = valid ? h : hend - 1;
   const long pset = (long)gene * a.K + iclass;
   const double *Pint = a.pint + pset * a.n_nodes * (N * N);
   const long tipstride = a.tip_words;
   const double *Ptip = a.ptip + pset * a.n_nodes * tipstride;

   double cur[N];
   double stk[MAXD][N];
   double lnscale = 0;
#pragma unroll
   for (int j = 0; j < N; j++) cur[j] = 0;

   for (int ip = 0;; ip++) {
      const Op op = fetch_op(a.ops, ip);
      if (op.code == OP_END) break;
      switch (op.code) {
      case OP_INIT_ONES: {
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] = 1.0;
      } break;
      case OP_INIT_TIP: {
         const int code = a.z[(long)op.a * a.z_stride + hc];
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] = (a.cleandata && j == code) ? 1.0 : 0.0;
      } break;
      case OP_MUL_TIP: {
         const int code = a.z[(long)op.a * a.z_stride + hc];
         const double *pt = Ptip + (long)op.a * tipstride + code * N;
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] *= pt[j];
      } break;
      case OP_SET_TIP: {
         const int code = a.z[(long)op.a * a.z_stride + hc];
         const double *pt = Ptip + (long)op.a * tipstride + code * N;
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] = pt[j];
      } break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2: {
         const int c1 = a.z[(long)op.a * a.z_stride + hc], c2 = a.z[(long)op.b * a.z_stride + hc];
         const double *p1 = Ptip + (long)op.a * tipstride + c1 * N;
         const double *p2 = Ptip + (long)op.b * tipstride + c2 * N;
         if (op.code == OP_SET_TIP2) {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] = p1[j] * p2[j];
         }
         else {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] = (cur[j] * p1[j]) * p2[j];
         }
      } break;
      case OP_PUSH: {
         if constexpr (REGSTK) {
#pragma unroll
            for (int d = 0; d < MAXD; d++)
               if (op.b == d) {
#pragma unroll
                  for (int j = 0; j < N; j++) stk[d][j] = cur[j];
               }
         }
         else {
#pragma unroll
            for (int j = 0; j < N; j++) stk[op.b][j] = cur[j];
         }
      } break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         const CONST_AS double *P = as_const(Pint + (long)op.a * (N * N));
         double out[N];
#pragma unroll
         for (int j = 0; j < N; j++) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < N; k++) t = fma(P[j * N + k], cur[k], t);
            out[j] = t;
         }
         const int pop = mm_pop_slot(op), push = mm_push_slot(op);
         if constexpr (REGSTK) {
#pragma unroll
            for (int d = 0; d < MAXD; d++)
               if (pop == d) {
#pragma unroll
                  for (int j = 0; j < N; j++) out[j] = stk[d][j] * out[j];
               }
#pragma unroll
            for (int d = 0; d < MAXD; d++)
               if (push == d) {
#pragma unroll
                  for (int j = 0; j < N; j++) stk[d][j] = out[j];
               }
         }
         else {
            if (pop >= 0) {
#pragma unroll
               for (int j = 0; j < N; j++) out[j] = stk[pop][j] * out[j];
            }
            if (push >= 0) {
#pragma unroll
               for (int j = 0; j < N; j++) stk[push][j] = out[j];
            }
         }
         if (push < 0) {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] = out[j];
         }
      } break;
      case OP_SCALE: {
         double mx = 0;
#pragma unroll
         for (int j = 0; j < N; j++) mx = cur[j] > mx ? cur[j] : mx;
         double fac;
         if (mx < 1e-300) {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] = 1.0;
            fac = -800;
         }
         else {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] /= mx;
            fac = log(mx);
         }
         lnscale += fac;
         if (a.keep && valid) a.scalef[((long)iclass * a.n_scale + op.b) * a.n_patt + h] = fac;
      } break;
      case OP_STORE: {
         if (valid) {
            double *dst = a.partials + (((long)iclass * a.n_int + (op.a - a.n_tips)) * a.n_patt + h) * N;
#pragma unroll
            for (int j = 0; j < N; j++) dst[j] = cur[j];
         }
      } break;
      case OP_LOAD: {
         const double *src = a.partials + (((long)iclass * a.n_int + (op.a - a.n_tips)) * a.n_patt + hc) * N;
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] = src[j];
      } break;
      case OP_EXPORT: {
         if (valid) {
            double *dst = a.export_buf + ((long)iclass * a.n_patt + h) * N;
#pragma unroll
            for (int j = 0; j < N; j++) dst[j] = cur[j];
            if (a.export_scale) a.export_scale[(long)iclass * a.n_patt + h] = lnscale;
         }
      } break;
      case OP_ROOT: {
         const double *pi = a.pi + (long)(a.n_pi > 1 ? gene : 0) * N;
         double f = 0;
#pragma unroll
         for (int j = 0; j < N; j++) f = fma(pi[j], cur[j], f);
         if (a.keep && a.n_scale) {
            lnscale = 0;
            if (valid)
               for (int k = 0; k < a.n_scale; k++) lnscale += a.scalef[((long)iclass * a.n_scale + k) * a.n_patt + h];
         }
         if (valid) {
            double out = 0;
            if (a.weights[h] > 0) out = root_value(a, f, lnscale);
            a.fhK[(long)iclass * a.n_patt + h] = out;
         }
      } break;
      default: break;
      }
   }
}

// ------------------------------------------------------------------------------------------------
// Reduction: per-pattern class mixture + log (lfundG treesub.c:7630-7657, lfun 7796-7800), then a
// fixed-order two-level sum of w_h * log f_h (deterministic for a given n_patt).
// ------------------------------------------------------------------------------------------------
struct ReduceArgs {
   double *fhK;        // [K][n_patt]; with `raw` it arrives as floored root sums and leaves as fx_r's values
   const double *weights, *freqK;
   const double *fscale; // raw + n_scale: summed scale factors [K][n_patt]
   int raw;
   long freqK_bs;        // batched evaluations: blockIdx.y = batch element; its classes, partial sums and output follow
                         // element 0's at strides K*n_patt, gridDim.x and 1; freqK at freqK_bs (0 = shared)
   double *lnf;        // optional [n_patt]
   double *partial;    // partial sums at their GLOBAL positions: element (batch b, chunk first_chunk + blockIdx.x) at
                       // b * nb_stride + first_chunk + blockIdx.x (one engine: first_chunk = 0, nb_stride = gridDim.x)
   double *out;        // scalar
   int n_patt, K, mode, n_scale, chunk;
   int first_chunk, nb_stride;
   int *counter;       // [batch] tickets of red_block_finish (null: the total is formed by reduce_stage2 after the all-reduce)
};

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

// Tile blocks of the specialised kernel (PruneArgs::ztiles): per 128-pattern tile the tip codes of its patterns, one
// 128-byte row per tip, then a row of weight > 0 flags; patterns past the tile's gene read as code 0 / flag 0.
__global__ __launch_bounds__(256) void ztile_kernel(const int2 *tiles, const int *gene_off, const unsigned char *z, long z_stride,
                                                    const double *weights, int n_tips, int zt_bytes, unsigned char *out)
{
   const int t = blockIdx.x, i = threadIdx.x, tp = blockDim.x;      // one thread per pattern of the tile (128 or 192)
   const int g = tiles[t].x, h = tiles[t].y + i, hend = gene_off[g + 1];
   unsigned char *o = out + (long)t * zt_bytes;
   for (int tip = 0; tip < n_tips; tip++) o[tip * tp + i] = h < hend ? z[tip * z_stride + h] : (unsigned char)0;
   o[n_tips * tp + i] = (h < hend && weights[h] > 0) ? 1 : 0;
   for (int k = (n_tips + 1) * tp + i; k < zt_bytes; k += tp) o[k] = 0;
}

// Tip codes pattern-major for the fused one-pattern-per-lane kernels: row h = zw dwords, byte t = code of tip t.
__global__ __launch_bounds__(256) void zpm_kernel(const unsigned char *z, long z_stride, int n_tips, int n_patt, int zw, unsigned int *out)
{
   const long h = (long)blockIdx.x * 256 + threadIdx.x;
   if (h >= n_patt) return;
   for (int w = 0; w < zw; w++) {
      unsigned int v = 0;
      for (int b = 0; b < 4; b++) {
         const int t = 4 * w + b;
         if (t < n_tips) v |= (unsigned int)z[t * z_stride + h] << (8 * b);
      }
      out[h * zw + w] = v;
   }
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
struct DerivArgs {
   int n, K, n_genes, n_labels, n_t, label, rate_gs;      // rate_gs: as PmatArgs
   const double *t;            // [n_t]
   const double *rate, *gene_rate, *qfactor;
   const int *eigen_of;
   const EigenDev *eigen;
   double *out;                // [pset][n_t][3][n*n]
   double *frag;               // non-null: also [pset][n_t][3][4096], each matrix in MFMA A-operand order (as pmat_kernel's pint)
};

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

struct BranchArgs {
   int n, K, n_genes, n_patt, n_t, n_pi, b_is_tip, n_codes;
   const double *A, *B;        // partial of class ir: A + ir * cls_stride, layout [n_patt][n] (the keep-partials layout of prune_valu)
   long cls_stride;
   const double *SA, *SB;      // scale factors: SA[(ir * n_scale + k) * n_patt + h] summed over the n_scale slots (SB unused) — null: none
   int n_scale;
   const unsigned char *zb;    // tip b: codes [n_patt]
   const int *n_chara;
   const unsigned char *chara_map;
   const double *pi, *freqK, *weights, *PdP;   // PdP: [pset][n_t][3][n*n]
   const int *gene_off;
   double *partial;            // [gridDim.x][n_t][3]
};

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
struct BranchMfmaArgs {
   int n, K, n_genes, n_patt, n_pi, n_tips, n_int, n_tiles, n_scale, n_t, it;
   int a_node, b_node;                 // the branch's two ends; b may be a tip (then its "partial" is the code's state set)
   const int2 *tiles;                  // 64-pattern tiles (gene, first pattern)
   const int *gene_off;
   const double *partials;             // [K][n_int][n_tiles * 4][1024]
   const double *scalef;               // [K][n_scale][n_patt] or null
   const unsigned char *zb;            // tip b: codes [n_patt]
   const unsigned long long *code_mask; // tip b: bit s set = state s belongs to the code
   const double *pi;                   // [n_pi][4][16]
   const double *freqK, *weights;
   const double *frag;                 // [pset][n_t][3][4096]
   double *partial;                    // [n_tiles][n_t][3] (this launch fills trial length `it`)
};

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
struct PostArgs {
   int n, K, n_genes, n_patt, n_pi;
   const double *L, *S;        // [K][n_patt][n], summed scale factors [K][n_patt] or null
   const double *pi, *freqK;
   const int *gene_off;
   double *post;               // [n_patt][n]
};

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

// ------------------------------------------------------------------------------------------------
// Bayes empirical Bayes grid integral (lfunNSsites_M2M8 codeml.c:6482-6580) over the class likelihoods of the last
// evaluation: n_grid parameter points, each a mixture of n_cls classes (proportion pcl[g][c], class index iw[g][c] into
// the K evaluated classes).  n_grid x n_patt x n_cls terms with a log each — 10^11 at 10^6 patterns.
//   beb_scale:   f[k][h] = fhK[k][h] / max_k fhK[k][h]                       (codeml.c:6297-6305); with scaling nodes fhK holds
//                log f + the scale factors and f[k][h] = exp(fhK[k][h] - max_k fhK[k][h])            (codeml.c:6286-6294)
//   beb_lnfx:    part[g][b] = sum over block b's patterns of w_h log sum_c pcl[g][c] f[iw[g][c]][h]
//   beb_finish:  lnfXs[g] = sum_b part[g][b] (fixed order);  fX = log sum_g exp(lnfXs[g]);  wg[g] = exp(lnfXs[g] - fX)
//   beb_post:    per pattern, sums over the grid of the class posteriors, omega and omega^2
// One pattern per lane with its K class values in registers; the grid tables are wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------------------
#define BEB_MAXK 32
#define BEB_MAXCLS 8
struct BebArgs {
   int n_patt, K, n_grid, n_cls, n_pblk, patt_per_blk;
   int log_form;             // fhK holds logarithms (trees with scaling nodes)
   const double *fhK, *weights;
   double *f;                // [K][n_patt] scaled copy
   const double *pcl;        // [n_grid][n_cls]
   const int *iw;            // [n_grid][n_cls]
   const double *w_class;    // [K]
   double *part;             // [n_grid][n_pblk]
   double *lnfxs, *wg, *fx;  // [n_grid], [n_grid], [1]
   double *pr_last, *mean_w, *sd_w;   // [n_patt]
};

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
      double s = 0;
      for (int b = 0; b < a.n_pblk; b++) s += a.part[(long)g * a.n_pblk + b];
      a.lnfxs[g] = s;
      mx = fmax(mx, s);
   }
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
