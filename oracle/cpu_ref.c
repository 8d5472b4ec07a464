/* oracle/cpu_ref.c — TEST INFRASTRUCTURE ONLY (see cpu_ref.h).
 *
 * Plain-C restatement of the PAML likelihood hot path: same loop nests, same summation order,
 * same numeric guards as the reference, written fresh against the behaviour of
 *   tools.c:516-546 (PMatUVRoot), tools.c:578-604 (PMatK80), baseml.c:1572-1589 (PMatCijk),
 *   codeml.c:3585-3595 (PMatJC69like), treesub.c:7503-7592 (GetPMatBranch),
 *   codeml.c:3526-3582 / baseml.c:1517-1570 (ConditionalPNode), treesub.c:7200-7230 (NodeScale),
 *   treesub.c:7696-7761 (fx_r), treesub.c:7608-7660 (lfundG), treesub.c:7764-7807 (lfun).
 * It is the checker for the HIP engine and the timed single-thread CPU baseline ("port");
 * it is never part of the product path.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "cpu_ref.h"

static __thread long g_npmat = 0;
long orc_last_npmat(void) { return g_npmat; }

/* tools.c:516-546.  P = I + sum_k U[:,k] expm1(t Root_k) V[k,:]; t<1e-100 -> I; entries <0 -> 0.
 * Accumulation order: k outer, i, j inner — as the reference. */
void orc_pmat_uvroot(double *P, double t, int n, const double *U, const double *V, const double *Root)
{
   int i, j, k;
   if (t < 1e-100) {
      memset(P, 0, (size_t)n * n * sizeof(double));
      for (i = 0; i < n; i++) P[i * n + i] = 1;
      return;
   }
   memset(P, 0, (size_t)n * n * sizeof(double));
   for (k = 0; k < n; k++) {
      double e = expm1(t * Root[k]);
      for (i = 0; i < n; i++) {
         double ue = U[i * n + k] * e;
         double *row = P + (size_t)i * n;
         const double *vr = V + (size_t)k * n;
         for (j = 0; j < n; j++) row[j] += ue * vr[j];
      }
   }
   for (i = 0; i < n; i++) P[i * n + i] += 1;
   for (i = 0; i < n * n; i++)
      if (P[i] < 0) P[i] = 0;
}

/* baseml.c:1572-1589.  P_ij = delta_ij + sum_{k>=1} Cijk[i][j][k] expm1(t Root_k); k=0 term has exptm1[0]=0.
 * No clamp and no small-t shortcut. */
void orc_pmat_cijk(double *P, double t, int n, int nR, const double *Cijk, const double *Root)
{
   int i, j, k;
   double e[64];
   e[0] = 0;
   for (k = 1; k < nR; k++) e[k] = expm1(t * Root[k]);
   for (i = 0; i < n; i++) {
      for (j = 0; j < n; j++) {
         double s = 0;
         for (k = 0; k < nR; k++) s += Cijk[(size_t)i * n * nR + j * nR + k] * e[k];
         P[i * n + j] = s;
      }
      P[i * n + i] += 1;
   }
}

/* tools.c:578-604 (state order T,C,A,G: transitions are T<->C and A<->G). */
void orc_pmat_k80(double *P, double t, double kappa)
{
   int i, j;
   double e1 = expm1(-4 * t / (kappa + 2)), e2;
   if (fabs(kappa - 1) < 1e-20) {
      for (i = 0; i < 4; i++)
         for (j = 0; j < 4; j++) P[i * 4 + j] = (i == j ? 1. + 3 / 4.0 * e1 : -e1 / 4);
      return;
   }
   e2 = expm1(-2 * t * (kappa + 1) / (kappa + 2));
   for (i = 0; i < 4; i++)
      for (j = 0; j < 4; j++) {
         if (i == j) P[i * 4 + j] = 1 + (e1 + 2 * e2) / 4;
         else if ((i ^ j) == 1) P[i * 4 + j] = (e1 - 2 * e2) / 4;   /* 0<->1, 2<->3 */
         else P[i * 4 + j] = -e1 / 4;
      }
}

/* UNREST: GetPMatBranch treesub.c:7524-7526 -> matexp(Q t, n, 7, 5), tools.c:4879-4921:
 * e^A = (I + A/m + (A/m)^2/2! + ... + (A/m)^7/7!)^m with m = 2^5. */
void orc_pmat_qmat(double *P, double t, int n, const double *Q)
{
   double B[64], T1[64], T2[64], S[64], factor = 1, *Tp = T1, *Tn = T2, *sw;
   int i, j, k, term, sq;
   for (i = 0; i < n * n; i++) { B[i] = T1[i] = Q[i] * t * (1.0 / 32); P[i] = B[i]; }
   for (i = 0; i < n; i++) P[i * n + i] += 1;
   for (term = 2; term <= 7; term++) {
      for (i = 0; i < n; i++)
         for (j = 0; j < n; j++) {
            double s = 0;
            for (k = 0; k < n; k++) s += Tp[i * n + k] * B[k * n + j];
            Tn[i * n + j] = s;
         }
      factor /= term;
      for (i = 0; i < n * n; i++) P[i] += Tn[i] * factor;
      sw = Tp; Tp = Tn; Tn = sw;
   }
   for (sq = 0; sq < 5; sq++) {
      for (i = 0; i < n; i++)
         for (j = 0; j < n; j++) {
            double s = 0;
            for (k = 0; k < n; k++) s += P[i * n + k] * P[k * n + j];
            S[i * n + j] = s;
         }
      for (i = 0; i < n * n; i++) P[i] = S[i];
   }
}

/* codeml.c:3585-3595 */
void orc_pmat_jc69like(double *P, double t, int n)
{
   int i;
   double pii = 1. / n + (1. - 1. / n) * exp(-n / (n - 1.) * t);
   double pij = (1. - pii) / (n - 1.);
   for (i = 0; i < n * n; i++) P[i] = pij;
   for (i = 0; i < n; i++) P[i * n + i] = pii;
}

/* Branch time (codeml.c:3547-3551) + eigen selection and Qfactor (treesub.c:7547-7588). */
void orc_pmat_branch(const orc_problem *pb, int gene, int iclass, int node, double *P)
{
   int n = pb->n, lab = pb->label ? pb->label[node] : 0;
   const orc_eigen *es = &pb->eigen[pb->eigen_of[((size_t)gene * pb->K + iclass) * pb->n_labels + lab]];
   double t = pb->branch[node] * pb->rate[gene * pb->rate_gs + iclass];
   t *= (pb->gene_rate ? pb->gene_rate[gene] : 1.0);
   g_npmat++;
   switch (es->kind) {
   case ORC_EIGEN_K80: orc_pmat_k80(P, t, es->kappa); break;
   case ORC_EIGEN_JC69LIKE: orc_pmat_jc69like(P, t, n); break;
   case ORC_EIGEN_QMAT: orc_pmat_qmat(P, t, n, es->U); break;      /* U carries Q */
   case ORC_EIGEN_CIJK: orc_pmat_cijk(P, t, n, es->nR, es->Cijk, es->Root); break;
   default:
      t *= (pb->qfactor ? pb->qfactor[(size_t)iclass * pb->n_labels + lab] : 1.0);
      orc_pmat_uvroot(P, t, n, es->U, es->V, es->Root);
   }
}

/* row stride of the tip codes: a pattern block of a larger data set keeps the parent's stride */
#define ZS(pb) ((pb)->z_stride ? (size_t)(pb)->z_stride : (size_t)(pb)->n_patt)

typedef struct {
   const orc_problem *pb;
   double *conP;      /* this class's slab: [n_nodes-n_tips][n_patt][n] */
   double *scalef;    /* this class's slab: [n_scale][n_patt] */
   double *PMat;
   double *tiproot;   /* [n_patt][n] partial of a root that is itself an observed sequence ("young ancestor") */
   int nthreads;
} ctx_t;

static double *node_conP(const ctx_t *c, int inode)
{
   if (inode < c->pb->n_tips) return c->tiproot;
   return c->conP + (size_t)(inode - c->pb->n_tips) * c->pb->n_patt * c->pb->n;
}

/* treesub.c:7200-7230 */
static void node_scale(const ctx_t *c, int inode, int pos0, int pos1)
{
   const orc_problem *pb = c->pb;
   int n = pb->n, j, k = 0, h;
   double *L = node_conP(c, inode);
   for (j = 0; j < pb->n_nodes; j++) {
      if (j == inode) break;
      if (pb->scale_node[j]) k++;
   }
   for (h = pos0; h < pos1; h++) {
      double t = 0;
      for (j = 0; j < n; j++)
         if (L[(size_t)h * n + j] > t) t = L[(size_t)h * n + j];
      if (t < 1e-300) {
         for (j = 0; j < n; j++) L[(size_t)h * n + j] = 1;
         c->scalef[(size_t)k * pb->n_patt + h] = -800;
      }
      else {
         for (j = 0; j < n; j++) L[(size_t)h * n + j] /= t;
         c->scalef[(size_t)k * pb->n_patt + h] = log(t);
      }
   }
}

/* codeml.c:3526-3582 / baseml.c:1517-1570 */
static void conditional_p_node(ctx_t *c, int inode, int igene, int iclass)
{
   const orc_problem *pb = c->pb;
   int n = pb->n, np = pb->n_patt, i, pos0 = pb->gene_off[igene], pos1 = pb->gene_off[igene + 1];
   int s0 = pb->sons_ptr[inode], s1 = pb->sons_ptr[inode + 1];
   double *L, *P = c->PMat;
   long h;

   for (i = s0; i < s1; i++) {
      int son = pb->sons[i];
      if (pb->sons_ptr[son + 1] > pb->sons_ptr[son]) conditional_p_node(c, son, igene, iclass);
   }
   L = node_conP(c, inode);
   if (inode < pb->n_tips) {   /* young ancestor: codeml.c:3535-3543 (indicator only when cleandata) */
      const unsigned char *z = pb->z + (size_t)inode * ZS(pb);
      for (h = (long)pos0 * n; h < (long)pos1 * n; h++) L[h] = 0;
      if (pb->cleandata)
         for (h = pos0; h < pos1; h++) L[h * n + z[h]] = 1;
   }
   else
      for (h = (long)pos0 * n; h < (long)pos1 * n; h++) L[h] = 1;

   for (i = s0; i < s1; i++) {
      int son = pb->sons[i];
      int son_is_tip = (pb->sons_ptr[son + 1] == pb->sons_ptr[son]);
      orc_pmat_branch(pb, igene, iclass, son, P);
      if (son_is_tip && pb->cleandata) {
         const unsigned char *z = pb->z + (size_t)son * ZS(pb);
#pragma omp parallel for if (c->nthreads > 1) num_threads(c->nthreads) schedule(static)
         for (h = pos0; h < pos1; h++) {
            int j;
            for (j = 0; j < n; j++) L[h * n + j] *= P[j * n + z[h]];
         }
      }
      else if (son_is_tip) {
         const unsigned char *z = pb->z + (size_t)son * ZS(pb);
#pragma omp parallel for if (c->nthreads > 1) num_threads(c->nthreads) schedule(static)
         for (h = pos0; h < pos1; h++) {
            int j, k, code = z[h], nc = pb->n_chara[code];
            const unsigned char *map = pb->chara_map + (size_t)code * n;
            for (j = 0; j < n; j++) {
               double t = 0;
               for (k = 0; k < nc; k++) t += P[j * n + map[k]];
               L[h * n + j] *= t;
            }
         }
      }
      else {
         const double *Ls = node_conP(c, son);
#pragma omp parallel for if (c->nthreads > 1) num_threads(c->nthreads) schedule(static)
         for (h = pos0; h < pos1; h++) {
            int j, k;
            for (j = 0; j < n; j++) {
               double t = 0;
               for (k = 0; k < n; k++) t += P[j * n + k] * Ls[h * n + k];
               L[h * n + j] *= t;
            }
         }
      }
   }
   if (pb->scale_node && pb->scale_node[inode]) node_scale(c, inode, pos0, pos1);
}

double orc_eval(const orc_problem *pb, double *lnf, double *fhK_out, double *partials, double *scalef_out, int nthreads)
{
   int n = pb->n, np = pb->n_patt, K = pb->K, nint = pb->n_nodes - pb->n_tips;
   int n_scale = 0, i, ir, ig, k;
   size_t slab = (size_t)nint * np * n;
   double *conP_own = NULL, *scalef_own = NULL, *fhK, lnL = 0;
   ctx_t c;
   long h;

   g_npmat = 0;
   if (pb->scale_node)
      for (i = 0; i < pb->n_nodes; i++) n_scale += (pb->scale_node[i] != 0);

   if (!partials) conP_own = (double *)malloc(slab * sizeof(double));
   if (!scalef_out && n_scale) scalef_own = (double *)calloc((size_t)n_scale * np, sizeof(double));
   fhK = fhK_out ? fhK_out : (double *)malloc((size_t)K * np * sizeof(double));
   c.pb = pb;
   c.PMat = (double *)malloc((size_t)n * n * sizeof(double));
   c.nthreads = nthreads > 1 ? nthreads : 1;
   c.tiproot = pb->root < pb->n_tips ? (double *)malloc((size_t)np * n * sizeof(double)) : NULL;

   /* fx_r (treesub.c:7713-7759) — also covers lfun, which is the K=1 case with its own floor */
   for (ig = 0; ig < pb->n_genes; ig++) {
      const double *pi = pb->pi + (size_t)(pb->n_pi > 1 ? ig : 0) * n;
      for (ir = 0; ir < K; ir++) {
         c.conP = partials ? partials + (size_t)ir * slab : conP_own;
         c.scalef = scalef_out ? scalef_out + (size_t)ir * n_scale * np : scalef_own;
         conditional_p_node(&c, pb->root, ig, ir);
         {
            const double *Lr = node_conP(&c, pb->root);
            for (h = pb->gene_off[ig]; h < pb->gene_off[ig + 1]; h++) {
               double fh = 0;
               if (pb->weights[h] <= 0) { fhK[(size_t)ir * np + h] = 0; continue; }
               for (i = 0; i < n; i++) fh += pi[i] * Lr[h * n + i];
               if (fh <= 0) fh = (pb->mode == ORC_MODE_LFUN ? 1e-80 : 1e-300);  /* treesub.c:7794 / 7741 */
               if (pb->mode == ORC_MODE_LFUN || n_scale) {
                  fh = log(fh);
                  for (k = 0; k < n_scale; k++) fh += c.scalef[(size_t)k * np + h];
               }
               fhK[(size_t)ir * np + h] = fh;
            }
         }
      }
   }

   /* lfun tail (treesub.c:7796-7803) or lfundG (treesub.c:7630-7657) */
   for (h = 0; h < np; h++) {
      double fh;
      if (pb->weights[h] <= 0) { if (lnf) lnf[h] = 0; continue; }
      if (pb->mode == ORC_MODE_LFUN)
         fh = fhK[h];
      else if (n_scale) {
         int it = 0;
         double t;
         for (ir = 1; ir < K; ir++)
            if (fhK[(size_t)ir * np + h] > fhK[(size_t)it * np + h]) it = ir;
         t = fhK[(size_t)it * np + h];
         for (ir = 0, fh = 0; ir < K; ir++) fh += pb->freqK[ir] * exp(fhK[(size_t)ir * np + h] - t);
         fh = t + log(fh);
      }
      else {
         for (ir = 0, fh = 0; ir < K; ir++) fh += pb->freqK[ir] * fhK[(size_t)ir * np + h];
         if (fh <= 0) fh = 1e-300;
         fh = log(fh);
      }
      lnL += fh * pb->weights[h];
      if (lnf) lnf[h] = fh;
   }

   free(c.PMat);
   free(c.tiproot);
   free(conP_own);
   free(scalef_own);
   if (!fhK_out) free(fhK);
   return lnL;
}


/* lfunAdG (treesub.c:7447-7494): auto-discrete-gamma rates.  fx_r gives fhK as for lfundG; the likelihood is then a
 * forward pass of the K-state rate chain with transition matrix MK over the sites in their original order
 * (pose[site] = pattern), rescaled at every site.  Returns +lnL. */
double orc_eval_adg(const orc_problem *pb, const double *MK, const int *pose, int ls)
{
   int K = pb->K, np = pb->n_patt, n_scale = 0, i, il, ir, j;
   double *fhK = (double *)malloc((size_t)K * np * sizeof(double)), *b1 = (double *)malloc(2 * K * sizeof(double)), *b2 = b1 + K;
   double lnL = 0, fh;
   long h;
   if (pb->scale_node)
      for (i = 0; i < pb->n_nodes; i++) n_scale += (pb->scale_node[i] != 0);
   orc_eval(pb, NULL, fhK, NULL, NULL, 1);
   if (n_scale)      /* treesub.c:7458-7464: class 0 carries the scale, the others become ratios to it */
      for (h = 0; h < np; h++) {
         fh = fhK[h];
         lnL += fh * pb->weights[h];
         fhK[h] = 1;
         for (ir = 1; ir < K; ir++) fhK[(size_t)ir * np + h] = exp(fhK[(size_t)ir * np + h] - fh);
      }
   for (il = 0; il < ls; il++) {
      h = pose[il];
      if (il == 0)
         for (ir = 0; ir < K; ir++) b1[ir] = fhK[(size_t)ir * np + h];
      else {
         for (ir = 0; ir < K; ir++) {
            for (j = 0, fh = 0; j < K; j++) fh += MK[ir * K + j] * b1[j];
            b2[ir] = fh * fhK[(size_t)ir * np + h];
         }
         for (ir = 0; ir < K; ir++) b1[ir] = b2[ir];
      }
      for (ir = 0, fh = 0; ir < K; ir++) fh += b1[ir];
      if (fh < 1e-90) fh = 1e-300;
      for (ir = 0; ir < K; ir++) b1[ir] /= fh;      /* abyx(1/fh, ...) */
      lnL += log(fh);
   }
   for (ir = 0, fh = 0; ir < K; ir++) fh += pb->freqK[ir] * b1[ir];
   lnL += log(fh);
   free(fhK); free(b1);
   return lnL;
}

/* The same evaluation with the patterns cut into blocks and the blocks spread over the host cores: every thread walks
 * the whole tree for its own block, so partials stay in its cache (the all-cores CPU baseline of bench.py; the
 * reference itself is single-threaded).  Single-gene problems.  Returns +lnL. */
double orc_eval_blocked(const orc_problem *pb, int nthreads, int block)
{
   int nb, b;
   double lnL = 0;
   if (pb->n_genes != 1 || block < 1) return 0.0 / 0.0;
   nb = (pb->n_patt + block - 1) / block;
#pragma omp parallel num_threads(nthreads > 1 ? nthreads : 1) reduction(+ : lnL)
   {
      /* one set of partial arrays per thread, reused for all its blocks (a fresh multi-megabyte malloc per block would
       * spend the time in page faults) */
      double *part = (double *)malloc((size_t)pb->K * (pb->n_nodes - pb->n_tips) * block * pb->n * sizeof(double));
#pragma omp for schedule(dynamic)
      for (b = 0; b < nb; b++) {
         orc_problem sub = *pb;
         int off[2];
         const int lo = b * block, len = lo + block <= pb->n_patt ? block : pb->n_patt - lo;
         off[0] = 0; off[1] = len;
         sub.n_patt = len;
         sub.z = pb->z + lo;
         sub.z_stride = (long)ZS(pb);
         sub.weights = pb->weights + lo;
         sub.gene_off = off;
         lnL += orc_eval(&sub, NULL, NULL, part, NULL, 1);
      }
      free(part);
   }
   return lnL;
}


/* ---------------------------------------------------------------------------------------------------------------
 * Branch-local lnL(t), dlnL/dt, d2lnL/dt2 (lfuntdd / lfuntdd_SiteClass, treesub.c:8204-8296, 8403-8541).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
   const orc_problem *pb;
   int *father;
   double *P;
} bctx_t;

/* message at x looking away from neighbour `from`: out[h][j], patterns [pos0,pos1) of gene ig, class ir */
static void msg(const bctx_t *c, int x, int from, int ig, int ir, double *out)
{
   const orc_problem *pb = c->pb;
   int n = pb->n, np = pb->n_patt, pos0 = pb->gene_off[ig], pos1 = pb->gene_off[ig + 1], i, j, k;
   long h;
   if (pb->sons_ptr[x + 1] == pb->sons_ptr[x] && x < pb->n_tips) {   /* tip: indicator of its state set */
      const unsigned char *z = pb->z + (size_t)x * ZS(pb);
      for (h = pos0; h < pos1; h++) {
         int code = z[h], nc = pb->n_chara[code];
         for (j = 0; j < n; j++) out[h * n + j] = 0;
         for (k = 0; k < nc; k++) out[h * n + pb->chara_map[(size_t)code * n + k]] = 1;
      }
      return;
   }
   for (h = (long)pos0 * n; h < (long)pos1 * n; h++) out[h] = 1;
   {
      /* neighbours: sons (edge stored at the son) and the father (edge stored at x) */
      int nn = pb->sons_ptr[x + 1] - pb->sons_ptr[x] + (c->father[x] >= 0 ? 1 : 0);
      double *tmp = (double *)malloc((size_t)np * n * sizeof(double));
      double *P = (double *)malloc((size_t)n * n * sizeof(double));
      for (i = 0; i < nn; i++) {
         int is_father = (i == nn - 1 && c->father[x] >= 0);
         int y = is_father ? c->father[x] : pb->sons[pb->sons_ptr[x] + i];
         int edge = is_father ? x : y;      /* node that carries this edge's branch length / label */
         if (y == from) continue;
         msg(c, y, x, ig, ir, tmp);
         orc_pmat_branch(pb, ig, ir, edge, P);
         for (h = pos0; h < pos1; h++)
            for (j = 0; j < n; j++) {
               double s = 0;
               for (k = 0; k < n; k++) s += P[j * n + k] * tmp[h * n + k];
               out[h * n + j] *= s;
            }
      }
      free(tmp);
      free(P);
   }
}

/* Marginal posterior of the states at an internal node (PostProbNode treesub.c:6142-6180 after ReRootTree at the node):
 * post[h][i] = sum_ir freqK_ir pi_i prod_{neighbours y of node} sum_j P_ij(t_y) M_y[h][j] / (sum over i).  The conditional
 * at the node is the product of the messages from all its neighbours, computed directly (no scaling: see above). */
int orc_node_posterior(const orc_problem *pb, int node, double *post)
{
   int n = pb->n, np = pb->n_patt, K = pb->K, i, ig, ir;
   long h;
   bctx_t c;
   double *L = (double *)malloc((size_t)np * n * sizeof(double));
   c.pb = pb;
   c.father = (int *)malloc(pb->n_nodes * sizeof(int));
   for (i = 0; i < pb->n_nodes; i++) c.father[i] = -1;
   for (i = 0; i < pb->n_nodes; i++) {
      int j;
      for (j = pb->sons_ptr[i]; j < pb->sons_ptr[i + 1]; j++) c.father[pb->sons[j]] = i;
   }
   for (h = 0; h < (long)np * n; h++) post[h] = 0;
   for (ir = 0; ir < K; ir++)
      for (ig = 0; ig < pb->n_genes; ig++) {
         const double *pi = pb->pi + (size_t)(pb->n_pi > 1 ? ig : 0) * n;
         msg(&c, node, -1, ig, ir, L);
         for (h = pb->gene_off[ig]; h < pb->gene_off[ig + 1]; h++)
            for (i = 0; i < n; i++) post[h * n + i] += pb->freqK[ir] * pi[i] * L[h * n + i];
      }
   for (h = 0; h < np; h++) {
      double s = 0;
      for (i = 0; i < n; i++) s += post[h * n + i];
      for (i = 0; i < n; i++) post[h * n + i] = s > 0 ? post[h * n + i] / s : 0;
   }
   free(L); free(c.father);
   return 0;
}

/* lfuntdd / lfuntdd_SiteClass restated (treesub.c:8204-8296, 8403-8541): P, dP, ddP from the eigen system, then the contraction with
 * the partials of the branch's two ends.  PINNING: the reference prints neither dl nor ddl, so this function is pinned (a) by l(t) = the
 * pinned lnL at the current length and by central differences of that pinned lnL along the branch (tests/test_oracle_golden.py), and
 * (b) through the reference itself: its own minbranches, run on the engine through integration/treesub_branch_seam.patch, follows the
 * Newton steps these derivatives dictate and ends at the unmodified program's printed optimum (HIV M0 / M2a, and -1035.530508 for
 * branch-site A under method = 1: tests/test_reference_binding_gpu.py), while the engine's values are held to this function's. */
int orc_eval_branch(const orc_problem *pb, int node_b, int n_t, const double *t, double *lnL, double *dlnL, double *ddlnL)
{
   int n = pb->n, np = pb->n_patt, K = pb->K, i, j, k, it, ig, ir, a;
   long h;
   bctx_t c;
   double *A, *B, *P, *dP, *ddP, *fh, *dfh, *ddfh;
   /* scale_node flags are not used here: node scaling multiplies a partial and divides it out again in the log, so it
    * changes no value, only the exponent range; the two messages below are computed unscaled (fine in double for the
    * tree sizes the tests use) */
   c.pb = pb;
   c.father = (int *)malloc(pb->n_nodes * sizeof(int));
   for (i = 0; i < pb->n_nodes; i++) c.father[i] = -1;
   for (i = 0; i < pb->n_nodes; i++)
      for (j = pb->sons_ptr[i]; j < pb->sons_ptr[i + 1]; j++) c.father[pb->sons[j]] = i;
   a = c.father[node_b];
   if (a < 0) { free(c.father); return -1; }
   A = (double *)malloc((size_t)K * np * n * sizeof(double));
   B = (double *)malloc((size_t)K * np * n * sizeof(double));
   P = (double *)malloc((size_t)3 * n * n * sizeof(double));
   dP = P + n * n; ddP = dP + n * n;
   fh = (double *)malloc((size_t)3 * np * sizeof(double));
   dfh = fh + np; ddfh = dfh + np;
   for (ir = 0; ir < K; ir++)
      for (ig = 0; ig < pb->n_genes; ig++) {
         msg(&c, a, node_b, ig, ir, A + (size_t)ir * np * n);
         msg(&c, node_b, a, ig, ir, B + (size_t)ir * np * n);
      }
   for (it = 0; it < n_t; it++) {
      double l = 0, dl = 0, ddl = 0;
      for (h = 0; h < np; h++) fh[h] = dfh[h] = ddfh[h] = 0;
      for (ir = 0; ir < K; ir++)
         for (ig = 0; ig < pb->n_genes; ig++) {
            int lab = pb->label ? pb->label[node_b] : 0;
            const orc_eigen *es = &pb->eigen[pb->eigen_of[((size_t)ig * K + ir) * pb->n_labels + lab]];
            const double *pi = pb->pi + (size_t)(pb->n_pi > 1 ? ig : 0) * n;
            int nroot = es->kind == ORC_EIGEN_CIJK ? es->nR : n;
            double qf = (es->kind == ORC_EIGEN_UVROOT && pb->qfactor) ? pb->qfactor[(size_t)ir * pb->n_labels + lab] : 1.0;
            for (i = 0; i < 3 * n * n; i++) P[i] = 0;
            if (es->kind == ORC_EIGEN_K80 || es->kind == ORC_EIGEN_JC69LIKE) {
               /* the reference reaches these models through Cijk of eigenTN93 (kappa1 = kappa2); the closed forms of
                * PMatK80 / PMatJC69like are the same functions of t:  P = 1/n + c1 e^{t mu1} + c2 e^{t mu2} */
               int k80 = es->kind == ORC_EIGEN_K80;
               double base = (pb->gene_rate ? pb->gene_rate[ig] : 1.0) * pb->rate[ig * pb->rate_gs + ir];
               double m1 = base * (k80 ? -4 / (es->kappa + 2) : -(double)n / (n - 1));
               double m2 = base * (k80 ? -2 * (es->kappa + 1) / (es->kappa + 2) : 0.0);
               double e1 = exp(t[it] * m1), e2 = k80 ? exp(t[it] * m2) : 0.0;
               for (i = 0; i < n; i++)
                  for (j = 0; j < n; j++) {
                     double c1, c2;
                     if (k80) { c1 = (i == j || (i ^ j) == 1) ? 0.25 : -0.25; c2 = i == j ? 0.5 : ((i ^ j) == 1 ? -0.5 : 0.0); }
                     else { c1 = i == j ? 1 - 1.0 / n : -1.0 / n; c2 = 0; }
                     P[i * n + j] = 1.0 / n + c1 * e1 + c2 * e2;
                     dP[i * n + j] = c1 * e1 * m1 + c2 * e2 * m2;
                     ddP[i * n + j] = c1 * e1 * m1 * m1 + c2 * e2 * m2 * m2;
                  }
               nroot = 0;
            }
            for (k = 0; k < nroot; k++) {
               /* treesub.c:8479-8483: multiply = rgene * Root[k] * _rateSite [* Qfactor_NS_branch] */
               double multiply = (pb->gene_rate ? pb->gene_rate[ig] : 1.0) * es->Root[k] * pb->rate[ig * pb->rate_gs + ir] * qf;
               double expt = k ? exp(t[it] * multiply) : 1.0;
               for (i = 0; i < n; i++)
                  for (j = 0; j < n; j++) {
                     double c0 = es->kind == ORC_EIGEN_CIJK ? es->Cijk[(size_t)i * n * nroot + j * nroot + k] * expt
                                                            : (es->U[i * n + k] * expt) * es->V[k * n + j];
                     P[i * n + j] += c0;
                     if (k) {
                        dP[i * n + j] += c0 * multiply;
                        ddP[i * n + j] += c0 * multiply * multiply;
                     }
                  }
            }
            for (h = pb->gene_off[ig]; h < pb->gene_off[ig + 1]; h++) {
               const double *Ah = A + ((size_t)ir * np + h) * n, *Bh = B + ((size_t)ir * np + h) * n;
               for (i = 0; i < n; i++) {
                  double piqi, pqj = 0, dpqj = 0, ddpqj = 0;
                  if (Bh[i] == 0) continue;
                  piqi = pb->freqK[ir] * pi[i] * Bh[i];
                  for (j = 0; j < n; j++) {
                     pqj += P[i * n + j] * Ah[j];
                     dpqj += dP[i * n + j] * Ah[j];
                     ddpqj += ddP[i * n + j] * Ah[j];
                  }
                  fh[h] += piqi * pqj;
                  dfh[h] += piqi * dpqj;
                  ddfh[h] += piqi * ddpqj;
               }
            }
         }
      for (h = 0; h < np; h++) {
         if (pb->weights[h] <= 0) continue;
         l += log(fh[h]) * pb->weights[h];
         dl += dfh[h] / fh[h] * pb->weights[h];
         ddl += (fh[h] * ddfh[h] - dfh[h] * dfh[h]) / (fh[h] * fh[h]) * pb->weights[h];
      }
      lnL[it] = l; dlnL[it] = dl; ddlnL[it] = ddl;
   }
   free(c.father); free(A); free(B); free(P); free(fh);
   return 0;
}
