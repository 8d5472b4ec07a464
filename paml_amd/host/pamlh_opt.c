/* pamlh_opt.c — maximum-likelihood estimation on top of the batched evaluation of the engine.
 *
 * The reference minimises -lnL with ming2 (tools.c:6595): BFGS, gradients by finite differences (gradientB tools.c:6561,
 * np or 2np calls of com.plfun per gradient, one after the other) and a line search of single evaluations (LineSearch2
 * tools.c:6279).  On the GPU one evaluation of a small data set is launch-latency bound, so the same mathematics is
 * arranged around paml_amd_eval_batch: all 2np central-difference points of a gradient are one launch, and all trial
 * step lengths of a line search are another.  Bounds follow SetxBound (codeml.c:1880, baseml.c:1100).  Written fresh;
 * only the published algorithm (BFGS with a box) is shared with the reference.
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pamlh_internal.h"

/* lnL at nb parameter vectors xs[nb][np] in one launch.  Vectors whose substitution-model part (x[ntime..np)) is equal
 * share one model set-up (eigen decompositions are the host's expensive part); a vector the model rejects
 * (e.g. class proportions summing above 1) gets lnL = -1e300. */
static int eval_batch_lnf_once(pamlh *p, int nb, const double *xs, double *lnL, double *lnf);
/* (a device eigen-decomposition that did not converge — PAML_AMD_ENOCONV from the evaluation behind it — sends the whole batch round again
 *  with the rate matrices decomposed on the host) */
static int eval_batch_lnf(pamlh *p, int nb, const double *xs, double *lnL, double *lnf)
{
   int rc = eval_batch_lnf_once(p, nb, xs, lnL, lnf);
   if (rc == PAML_AMD_ENOCONV && pamlh_force_host_eigen()) rc = eval_batch_lnf_once(p, nb, xs, lnL, lnf);
   return rc;
}

int pamlh_eval_batch_gpu(pamlh *p, int nb, const double *xs, double *lnL) { return eval_batch_lnf(p, nb, xs, lnL, NULL); }

/* ... with the per-pattern log f_h of every vector, lnf[nb][npatt], when lnf is not NULL.
 * The model set-ups of the distinct substitution-parameter vectors (eigen decompositions: the host's expensive part, one per
 * site class) are independent of each other and run on all host cores, each on its own copy of the model state. */
/* PAMLH_TIMING=1: where a search's wall time goes, by phase of the batched evaluation, printed when the process ends. */
static double tm_acc[5];
static long tm_n;
static int tm_on = -1;
static double tm_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void tm_report(void)
{
   fprintf(stderr, "pamlh timing: %ld batched evaluations; model set-ups %.1f ms, collecting eigen systems %.1f, eigen call %.1f, tables %.1f, "
           "set_classes + set_pi + eval_batch (P(t), pruning, the wait) %.1f\n", tm_n, tm_acc[0] * 1e3, tm_acc[1] * 1e3, tm_acc[2] * 1e3, tm_acc[3] * 1e3, tm_acc[4] * 1e3);
}
#define TM(k) do { if (tm_on > 0) { const double t_ = tm_now(); tm_acc[k] += t_ - tm_t; tm_t = t_; } } while (0)

static int eval_batch_lnf_once(pamlh *p, int nb, const double *xs, double *lnL, double *lnf)
{
   double tm_t = 0;
   const int np = p->np, nt = p->ntime, nm = np - nt, nn = p->nnode;
   int *cand_of = (int *)malloc(nb * sizeof(int)), *cand_elem = (int *)malloc(nb * sizeof(int)), *cand_rep = (int *)malloc(nb * sizeof(int));
   int ncand = 0, nrep = 0, b, c, i, rc = 0, K = 0, L = 1, n_eigen = 0, mode = 0, n_pi = 1, RK = 0;
   pamlh **ws = (pamlh **)calloc(nb, sizeof(pamlh *));
   double *br = (double *)calloc((size_t)nb * nn, sizeof(double)), *fk = NULL, *rt = NULL, *rep_fk = NULL, *rep_rt = NULL;
   const double *pi = NULL;
   int *eo = NULL, *rep_eo = NULL, use_qf = 0;
   double *qf = NULL, *rep_qf = NULL, *gr = NULL;
   const int G = p->ngene;
   pamlh_eig_batch ebatch;
   memset(&ebatch, 0, sizeof(ebatch));
   if ((rc = pamlh_engine_ready(p))) goto done;
   if (tm_on < 0) { tm_on = getenv("PAMLH_TIMING") != NULL; if (tm_on) atexit(tm_report); }
   if (tm_on > 0) { tm_t = tm_now(); tm_n++; }
   if (!p->fix_rho || p->rho0 != 0) {      /* lfunAdG ends in a sequential chain over the sites on the host: one evaluation at a time */
      for (b = 0; b < nb; b++) {
         if (pamlh_set_x(p, xs + (size_t)b * np, np) || !pamlh_model_feasible(p)) { lnL[b] = -1e300; continue; }
         if ((rc = pamlh_eval_gpu(p, &lnL[b], lnf ? lnf + (size_t)b * p->npatt : NULL))) goto done;
      }
      goto done;
   }
   for (b = 0; b < nb; b++) {          /* distinct model parts */
      const double *x = xs + (size_t)b * np;
      for (c = 0; c < ncand; c++)
         if (!nm || !memcmp(x + nt, xs + (size_t)cand_elem[c] * np + nt, nm * sizeof(double))) break;
      if (c == ncand) cand_elem[ncand++] = b;
      cand_of[b] = c;
   }
   /* (a team no larger than the work: waking every core of a large host for a dozen items costs more than it saves) */
#pragma omp parallel for schedule(dynamic) num_threads(ncand < 16 ? ncand : 16) if (ncand > 2)
   for (c = 0; c < ncand; c++) {
      pamlh *q = pamlh_state_clone(p);
      if (q && (pamlh_set_x(q, xs + (size_t)cand_elem[c] * np, np) || !pamlh_model_feasible(q))) { pamlh_state_free(q); q = NULL; }
      ws[c] = q;
   }
   TM(0);
   for (c = 0; c < ncand; c++) {       /* accepted set-ups become the batch's eigen sets and class tables, in order */
      pamlh *q = ws[c];
      cand_rep[c] = -1;
      if (!q) continue;
      if (!nrep) {
         K = q->K; L = q->n_labels * G; n_eigen = q->n_eigen; mode = q->mode; pi = q->pi; n_pi = q->n_pi;
         rep_fk = (double *)malloc((size_t)ncand * K * sizeof(double));
         RK = (p->malpha ? G : 1) * K;
         rep_rt = (double *)malloc((size_t)ncand * RK * sizeof(double));
         rep_eo = (int *)malloc((size_t)ncand * K * L * sizeof(int));
         rep_qf = (double *)malloc((size_t)ncand * K * L * sizeof(double));
         use_qf = q->use_qf;
      }
      else if (q->K != K || q->n_labels * G != L || q->n_eigen != n_eigen || q->mode != mode) { rc = pamlh_fail(p, "internal: model shape changed inside a batch"); goto done; }
      if ((nrep + 1) * n_eigen > 4096) { rc = pamlh_fail(p, "batch needs more than 4096 eigen systems"); goto done; }
      /* (reversible rate matrices are only collected here: all the candidates' decompositions are ONE device call below) */
      if ((rc = pamlh_upload_eigen_sets(q, p->eng, nrep * n_eigen, &ebatch))) { pamlh_fail(p, "%s", pamlh_error(q)); goto done; }
      memcpy(rep_fk + (size_t)nrep * K, q->freqK, K * sizeof(double));
      memcpy(rep_rt + (size_t)nrep * RK, q->rate, RK * sizeof(double));
      for (i = 0; i < K * L; i++) rep_eo[(size_t)nrep * K * L + i] = nrep * n_eigen + (G > 1 ? q->gene_eigen_of[i] : q->eigen_of[i]);
      for (i = 0; i < K * L; i++) rep_qf[(size_t)nrep * K * L + i] = q->use_qf ? q->qfactor[i] : 1.0;
      cand_rep[c] = nrep++;
   }
   if (!nrep) { for (b = 0; b < nb; b++) lnL[b] = -1e300; goto done; }
   TM(1);
   if ((rc = pamlh_eig_batch_flush(p, p->eng, &ebatch))) goto done;
   TM(2);
   fk = (double *)malloc((size_t)nb * K * sizeof(double));
   rt = (double *)malloc((size_t)nb * RK * sizeof(double));
   eo = (int *)malloc((size_t)nb * K * L * sizeof(int));
   qf = (double *)malloc((size_t)nb * K * L * sizeof(double));
   if (G > 1) gr = (double *)malloc((size_t)nb * G * sizeof(double));
   for (b = 0; b < nb; b++) {
      const double *x = xs + (size_t)b * np;
      const int cb = cand_of[b] < 0 ? -1 - cand_of[b] : cand_of[b];
      const int r = cand_rep[cb] < 0 ? 0 : cand_rep[cb];
      memcpy(fk + (size_t)b * K, rep_fk + (size_t)r * K, K * sizeof(double));
      memcpy(rt + (size_t)b * RK, rep_rt + (size_t)r * RK, RK * sizeof(double));
      memcpy(eo + (size_t)b * K * L, rep_eo + (size_t)r * K * L, (size_t)K * L * sizeof(int));
      memcpy(qf + (size_t)b * K * L, rep_qf + (size_t)r * K * L, (size_t)K * L * sizeof(double));
      if (pamlh_x_to_branches(p, x, br + (size_t)b * nn)) {      /* (clock: a node older than its ancestor) */
         for (i = 0; i < nn; i++) br[(size_t)b * nn + i] = 0.1;
         cand_of[b] = -1 - cand_of[b];
      }
      if (gr) { gr[(size_t)b * G] = 1; for (i = 1; i < G; i++) gr[(size_t)b * G + i] = x[nt + i - 1]; }
   }
   TM(3);
   /* (with several genes the tables are [gene][class]: one label, and L counts the genes) */
   if ((rc = paml_amd_set_classes(p->eng, mode, K, rep_fk, rep_rt, G > 1 ? 1 : L, rep_eo, use_qf ? rep_qf : NULL)) ||
       (p->malpha && (rc = paml_amd_set_gene_class_rates(p->eng, rep_rt)))) { rc = pamlh_fail(p, "%s", paml_amd_last_error(p->eng)); goto done; }
   {
      /* The engine evaluates a batch under ONE set of root frequencies (paml_amd_set_pi).  Where the frequencies are parameters
       * (nhomo = 1, 3, 4: com.pi is part of x) the elements are evaluated in groups of equal frequencies: a gradient batch is one
       * group for all the branch-length elements and one small group per perturbed model parameter. */
      const size_t pib = (size_t)n_pi * p->n * sizeof(double);
      int *grp = (int *)malloc(nb * sizeof(int)), ngrp = 0, gsel;
      const double **gpi = (const double **)malloc(nb * sizeof(double *));
      for (b = 0; b < nb; b++) {
         const int cb = cand_of[b] < 0 ? -1 - cand_of[b] : cand_of[b];
         const double *mypi = ws[cb] ? ws[cb]->pi : pi;
         for (gsel = 0; gsel < ngrp; gsel++) if (!memcmp(gpi[gsel], mypi, pib)) break;
         if (gsel == ngrp) gpi[ngrp++] = mypi;
         grp[b] = gsel;
      }
      if (ngrp == 1) {
         if ((rc = paml_amd_set_pi(p->eng, n_pi, gpi[0])) || (rc = paml_amd_eval_batch(p->eng, nb, br, gr, eo, use_qf ? qf : NULL, fk, rt, lnL, lnf))) {
            const int code = rc;
            rc = pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
            if (code == PAML_AMD_ENOCONV) rc = code;
         }
      }
      else {
         double *sbr = (double *)malloc((size_t)nb * nn * sizeof(double)), *sfk = (double *)malloc((size_t)nb * K * sizeof(double)), *srt = (double *)malloc((size_t)nb * RK * sizeof(double));
         double *sqf = (double *)malloc((size_t)nb * K * L * sizeof(double)), *sgr = gr ? (double *)malloc((size_t)nb * G * sizeof(double)) : NULL;
         double *sl = (double *)malloc(nb * sizeof(double)), *slf = lnf ? (double *)malloc((size_t)nb * p->npatt * sizeof(double)) : NULL;
         int *seo = (int *)malloc((size_t)nb * K * L * sizeof(int)), *idx = (int *)malloc(nb * sizeof(int));
         for (gsel = 0; gsel < ngrp && !rc; gsel++) {
            int m = 0;
            for (b = 0; b < nb; b++) {
               if (grp[b] != gsel) continue;
               memcpy(sbr + (size_t)m * nn, br + (size_t)b * nn, nn * sizeof(double));
               memcpy(sfk + (size_t)m * K, fk + (size_t)b * K, K * sizeof(double));
               memcpy(srt + (size_t)m * RK, rt + (size_t)b * RK, RK * sizeof(double));
               memcpy(sqf + (size_t)m * K * L, qf + (size_t)b * K * L, (size_t)K * L * sizeof(double));
               memcpy(seo + (size_t)m * K * L, eo + (size_t)b * K * L, (size_t)K * L * sizeof(int));
               if (sgr) memcpy(sgr + (size_t)m * G, gr + (size_t)b * G, G * sizeof(double));
               idx[m++] = b;
            }
            if ((rc = paml_amd_set_pi(p->eng, n_pi, gpi[gsel])) || (rc = paml_amd_eval_batch(p->eng, m, sbr, sgr, seo, use_qf ? sqf : NULL, sfk, srt, sl, slf))) {
               const int code = rc;
               rc = pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
               if (code == PAML_AMD_ENOCONV) rc = code;
               break;
            }
            for (i = 0; i < m; i++) {
               lnL[idx[i]] = sl[i];
               if (lnf) memcpy(lnf + (size_t)idx[i] * p->npatt, slf + (size_t)i * p->npatt, p->npatt * sizeof(double));
            }
         }
         free(sbr); free(sfk); free(srt); free(sqf); free(sgr); free(sl); free(slf); free(seo); free(idx);
      }
      free(grp); free(gpi);
      TM(4);
      if (rc) goto done;
   }
   for (b = 0; b < nb; b++)
      if (cand_of[b] < 0 || cand_rep[cand_of[b]] < 0 || !(lnL[b] == lnL[b])) lnL[b] = -1e300;
done:
   free(ebatch.ids); free(ebatch.Q); free(ebatch.pi); free(ebatch.scale);
   for (c = 0; c < ncand; c++) pamlh_state_free(ws[c]);
   free(ws); free(cand_of); free(cand_elem); free(cand_rep); free(br); free(fk); free(rt); free(eo); free(rep_fk); free(rep_rt); free(rep_eo); free(qf); free(rep_qf); free(gr);
   return rc;
}

/* Global clock: the optimiser works on y = (root age, age / father's age for the other internal nodes), where the ordering of
 * the ages is a box (the reference's own transformation while iterating, SetAge treesub.c:3794-3797); x holds the ages. */
static void clock_rec_y_to_x(const pamlh *p, int node, const double *y, double *x)
{
   int j;
   if (node < p->ns) return;
   if (node == p->root) x[node - p->ns] = y[node - p->ns];
   else {      /* between the lowest possible age (0, or the oldest dated tip below) and the father's (SetAge treesub.c:3723-3730) */
      const double low = p->tipdate ? p->age_low[node] : 0;
      x[node - p->ns] = low + (x[p->father[node] - p->ns] - low) * y[node - p->ns];
   }
   for (j = p->sons_ptr[node]; j < p->sons_ptr[node + 1]; j++) clock_rec_y_to_x(p, p->sons[j], y, x);
}

/* Proportions that sum to at most 1 (site-class proportions, base-frequency sets: k free values, the last category takes the
 * rest) are iterated on as y_i = log(p_i / p_last), p_i = e^{y_i} / (1 + sum_j e^{y_j}) — the reference's own transformation while
 * it iterates (f_and_x tools.c:2016, LASTROUND = 0): no point of the box is infeasible and a class that vanishes at the maximum
 * is a bound of the box.  Groups as (first index, k) in x. */
#define Y_SIMPLEX 30.0
int pamlh_simplex_groups(const pamlh *p, int *start, int *len, int cap)
{
   int n = 0, j;
   const int k0 = p->ntime + (p->ngene - 1);
   if (p->ngene > 1) return 0;
   if (p->seqtype == 1) {
      const int k = k0 + !p->fix_kappa + p->npi;      /* (x: kappa, the npi codon-frequency parameters, then the site-class proportions: pamlh_set_x) */
      int m = 0;
      if (p->aadist == 7) m = 0;
      else if (p->model >= 2 && p->nssites) m = 2;
      else if (p->model == 0 && (p->nssites == 2 || p->nssites == 12 || p->nssites == 13)) m = 2;
      else if (p->model == 0 && p->nssites == 3) m = p->ncatG - 1;
      else if (p->model == 0 && p->nssites == 4) m = 4;
      if (m >= 2 && n < cap) { start[n] = k; len[n++] = m; }
   }
   else if (p->seqtype == 0 && p->model != T92) {
      if (p->nhomo == 1 && n < cap) {
         const int nk = ((p->model == K80 || p->model == HKY85 || p->model == F84) && !p->fix_kappa) ? 1 : (p->model == TN93 && !p->fix_kappa) ? 2 : p->model == REV ? 5 : 0;
         start[n] = k0 + nk; len[n++] = 3;
      }
      if (p->nhomo > 2)
         for (j = 0; j < pamlh_nh_npi(p) && n < cap; j++) { start[n] = p->ntime + pamlh_nh_nrate(p) + 3 * j; len[n++] = 3; }
   }
   return n;
}

static void simplex_y_to_x(const pamlh *p, double *x)      /* in place */
{
   int start[PAMLH_MAXEIG + 4], len[PAMLH_MAXEIG + 4], n = pamlh_simplex_groups(p, start, len, PAMLH_MAXEIG + 4), g, i;
   for (g = 0; g < n; g++) {
      double *v = x + start[g], t = 1, mx = 0;
      for (i = 0; i < len[g]; i++) if (v[i] > mx) mx = v[i];
      t = exp(-mx);
      for (i = 0; i < len[g]; i++) { v[i] = exp(v[i] - mx); t += v[i]; }
      for (i = 0; i < len[g]; i++) v[i] /= t;
   }
}

static void simplex_x_to_y(const pamlh *p, double *x)      /* in place */
{
   int start[PAMLH_MAXEIG + 4], len[PAMLH_MAXEIG + 4], n = pamlh_simplex_groups(p, start, len, PAMLH_MAXEIG + 4), g, i;
   for (g = 0; g < n; g++) {
      double *v = x + start[g], last = 1;
      for (i = 0; i < len[g]; i++) last -= v[i];
      if (last < 1e-15) last = 1e-15;      /* (the difference of doubles near 1 resolves no less) */
      for (i = 0; i < len[g]; i++) {
         const double y = log((v[i] > 1e-300 ? v[i] : 1e-300) / last);
         v[i] = y < -Y_SIMPLEX ? -Y_SIMPLEX : y > Y_SIMPLEX ? Y_SIMPLEX : y;
      }
   }
}

static void clock_y_to_x(const pamlh *p, const double *y, double *x)
{
   memcpy(x, y, p->np * sizeof(double));
   if (p->clock) clock_rec_y_to_x(p, p->root, y, x);
   simplex_y_to_x(p, x);
}

static void clock_x_to_y(const pamlh *p, const double *x, double *y)
{
   int node;
   memcpy(y, x, p->np * sizeof(double));
   simplex_x_to_y(p, y);
   if (!p->clock) return;
   for (node = p->ns; node < p->nnode; node++)
      if (node != p->root) {
         const double low = p->tipdate ? p->age_low[node] : 0, fa = x[p->father[node] - p->ns] - low;
         y[node - p->ns] = fa > 0 ? (x[node - p->ns] - low) / fa : 0;
      }
}

/* lnL at nb vectors in the optimiser's variables */
static int batch_eval(pamlh *p, int nb, const double *ys, double *lnL)
{
   double *xs;
   int b, rc;
   if (!p->clock && !p->opt_transformed) return pamlh_eval_batch_gpu(p, nb, ys, lnL);
   xs = (double *)malloc((size_t)nb * p->np * sizeof(double));
   for (b = 0; b < nb; b++) clock_y_to_x(p, ys + (size_t)b * p->np, xs + (size_t)b * p->np);
   rc = pamlh_eval_batch_gpu(p, nb, xs, lnL);
   free(xs);
   return rc;
}

/* Box for x[] (SetxBound codeml.c:1880-1960, baseml.c:1100-1150): branch lengths [4e-6, 50], kappa and omega
 * [1e-4, 999], proportions (0, 1), beta and gamma shape parameters [0.005, 99], REV rates [1e-4, 999]. */
int pamlh_bounds(const pamlh *p, double *lo, double *hi)
{
   int k = 0, i, g;
   const int rep = (p->ngene > 1 && p->mgene >= 3) ? p->ngene : 1;
   for (i = 0; i < p->ntime; i++) { lo[k] = p->clock ? (i >= p->ns - 1 ? 1e-4 : 0) : p->fix_blength == 3 ? 1e-4 : 4e-6; hi[k++] = (p->clock && i >= p->ns - 1) ? 99 : 50; }
   for (i = 1; i < p->ngene; i++) { lo[k] = p->is_codeml ? 0.01 : 1e-4; hi[k++] = p->is_codeml ? 99 : 999; }      /* rgene (SetxBound) */
   for (g = 0; g < rep; g++)
   if (p->seqtype == 1) {
      if (!p->fix_kappa) { lo[k] = 1e-4; hi[k++] = 999; }
      for (i = 0; i < p->npi; i++) {      /* frequency ratios, then log fitnesses (SetxBound codeml.c:1921-1928) */
         const int nratio = (p->mutsel || p->codonfreq == 1) ? 3 : p->codonfreq == 2 ? 9 : 0;
         lo[k] = i < nratio ? 1e-4 : -29; hi[k++] = i < nratio ? 999 : 29;
      }
      if (p->aadist == 7) { for (i = 0; i < p->n_omega_type * (p->model == 2 ? p->n_omega : 1); i++) { lo[k] = 1e-4; hi[k++] = 999; } }
      else if (p->aadist) { lo[k] = 1e-4; hi[k++] = p->aadist < 0 ? 1 : 999; lo[k] = 1e-4; hi[k++] = 999; }
      else if (p->nssites == 0 && p->model == 2) { for (i = 0; i < p->n_omega - (p->fix_omega != 0); i++) { lo[k] = 1e-4; hi[k++] = 999; } }
      else if (p->model >= 2) {      /* branch-site A / B, clade C / D (SetxBound codeml.c:1940-1965; proportions untransformed here) */
         lo[k] = 1e-6; hi[k++] = 1 - 1e-6; lo[k] = 1e-6; hi[k++] = 1 - 1e-6;
         if (p->model == 2 && p->nssites == 2) { lo[k] = 1e-6; hi[k++] = 1; if (!p->fix_omega) { lo[k] = 1; hi[k++] = 999; } }
         else if (p->model == 2) { for (i = 0; i < 3; i++) { lo[k] = 1e-6; hi[k++] = 999; } }
         else {
            if (p->nssites == 2) { lo[k] = 1e-6; hi[k++] = 1; }
            else { lo[k] = 1e-4; hi[k++] = 1; lo[k] = 0.01; hi[k++] = 1.5; }
            for (i = 0; i < p->n_omega - (p->fix_omega != 0); i++) { lo[k] = 1e-6; hi[k++] = 999; }
         }
      }
      else if (p->nssites == 4) { for (i = 0; i < 4; i++) { lo[k] = 1e-6; hi[k++] = 1 - 1e-6; } }
      else if (p->nssites == 5) { lo[k] = 0.005; hi[k++] = 99; lo[k] = 0.005; hi[k++] = 99; }
      else if (p->nssites == 6) { lo[k] = 1e-6; hi[k++] = 1 - 1e-6; for (i = 0; i < 3; i++) { lo[k] = 0.005; hi[k++] = 99; } }
      else if (p->nssites == 9 || p->nssites == 10) { lo[k] = 1e-6; hi[k++] = 1 - 1e-6; for (i = 0; i < 4; i++) { lo[k] = 0.005; hi[k++] = 99; } }
      else if (p->nssites == 11) { lo[k] = 1e-6; hi[k++] = 1 - 1e-6; lo[k] = 0.005; hi[k++] = 99; lo[k] = 0.005; hi[k++] = 99; lo[k] = 1; hi[k++] = 9; lo[k] = 0.005; hi[k++] = 99; }
      else if (p->nssites == 12) { for (i = 0; i < 2; i++) { lo[k] = 1e-6; hi[k++] = 1 - 1e-6; } for (i = 0; i < 3; i++) { lo[k] = 1e-4; hi[k++] = 29; } }
      else if (p->nssites == 13) { for (i = 0; i < 2; i++) { lo[k] = 1e-6; hi[k++] = 1 - 1e-6; } for (i = 0; i < 4; i++) { lo[k] = 1e-4; hi[k++] = 29; } }
      else if (p->nssites == 3) {
         for (i = 0; i < p->ncatG - 1; i++) { lo[k] = 1e-6; hi[k++] = 1 - 1e-6; }
         for (i = 0; i < p->ncatG; i++) { lo[k] = 1e-6; hi[k++] = 999; }
      }
      else if (p->nssites == 0) { if (!p->fix_omega) { lo[k] = 1e-4; hi[k++] = 999; } }
      else if (p->nssites == 1) { lo[k] = 1e-6; hi[k++] = 1 - 1e-6; lo[k] = 1e-6; hi[k++] = 1; }
      else if (p->nssites == 2) {
         lo[k] = 1e-6; hi[k++] = 1 - 1e-6; lo[k] = 1e-6; hi[k++] = 1 - 1e-6;
         lo[k] = 1e-6; hi[k++] = 1; if (!p->fix_omega) { lo[k] = p->m2a_rel ? 1e-6 : 1; hi[k++] = 999; }
      }
      else if (p->nssites == 7) { lo[k] = 0.005; hi[k++] = 99; lo[k] = 0.005; hi[k++] = 99; }
      else if (p->nssites == 8) {
         lo[k] = 1e-6; hi[k++] = 1 - 1e-6; lo[k] = 0.005; hi[k++] = 99; lo[k] = 0.005; hi[k++] = 99;
         if (!p->fix_omega) { lo[k] = 1; hi[k++] = 999; }
      }
   }
   else if (p->seqtype == 2) {
      if (p->aa_model == 6 && !p->fix_kappa) { lo[k] = 1e-4; hi[k++] = 999; }
      if (p->aa_model >= 8) for (i = 0; i < p->n_aarate; i++) { lo[k] = 1e-5; hi[k++] = 999; }
   }
   else if (p->seqtype == 0) {
      const int nk = ((p->model == K80 || p->model == HKY85 || p->model == F84 || p->model == T92) && !p->fix_kappa) ? 1 : (p->model == TN93 && !p->fix_kappa) ? 2 : p->model == REV ? 5 : p->model == UNREST ? 11 : 0;
      for (i = 0; i < nk; i++) { lo[k] = 1e-4; hi[k++] = 999; }
   }
   if (p->seqtype == 0 && p->nhomo == 1) for (i = 0; i < (p->model == T92 ? 1 : 3); i++) { lo[k] = 1e-5; hi[k++] = 0.99999; }
   if (p->seqtype == 0 && p->nhomo >= 2) {      /* replaces the homogeneous model's rate parameters */
      k = p->ntime;
      for (i = 0; i < pamlh_nh_nrate(p); i++) { lo[k] = 1e-4; hi[k++] = 999; }
      for (i = 0; i < (p->nhomo > 2 ? pamlh_nh_npi(p) * (p->model == T92 ? 1 : 3) : 0); i++) { lo[k] = 1e-5; hi[k++] = 0.99999; }
   }
   if (!p->fix_alpha && !(p->seqtype == 1 && p->nssites)) for (i = 0; i < (p->malpha ? p->ngene : 1); i++) { lo[k] = 0.005; hi[k++] = 99; }
   if (!p->fix_rho) { lo[k] = -0.2; hi[k++] = 0.99; }
   return k == p->np ? 0 : -1;
}

static double dot(const double *a, const double *b, int n)
{
   double s = 0;
   int i;
   for (i = 0; i < n; i++) s += a[i] * b[i];
   return s;
}

/* Central-difference gradient of f = -lnL at x (one batch of <= 2 np points; one-sided where the box is in the way, or where
 * the model rejects the nudged vector — class proportions are iterated untransformed, so next to p0 + p1 = 1 a side can be
 * infeasible and comes back as the -1e300 marker: it is a blocked side, not a value).  Frozen parameters are not nudged. */
static int gradient(pamlh *p, const double *x, double f0, const double *lo, const double *hi, double *g, double *xs, double *ls, int *n_eval)
{
   const int n = p->np;
   int i, rc, na = 0;
   int *act = (int *)malloc((n + 1) * sizeof(int));
   for (i = 0; i < n; i++) {
      const double h = 1e-6 * (fabs(x[i]) + 1);
      double *xp, *xm;
      g[i] = 0;
      if (p->frozen && p->frozen[i]) continue;
      xp = xs + (size_t)(2 * na) * n; xm = xp + n;
      memcpy(xp, x, n * sizeof(double));
      memcpy(xm, x, n * sizeof(double));
      if (x[i] + h <= hi[i]) xp[i] = x[i] + h;
      if (x[i] - h >= lo[i]) xm[i] = x[i] - h;
      act[na++] = i;
   }
   if (na && (rc = batch_eval(p, 2 * na, xs, ls))) { free(act); return rc; }
   *n_eval += 2 * na;
   for (int a = 0; a < na; a++) {
      const double *xp = xs + (size_t)(2 * a) * n, *xm = xp + n;
      i = act[a];
      {
         const int okp = xp[i] != x[i] && ls[2 * a] > -1e299, okm = xm[i] != x[i] && ls[2 * a + 1] > -1e299;
         const double fp = okp ? -ls[2 * a] : f0, fm = okm ? -ls[2 * a + 1] : f0;
         const double vp = okp ? xp[i] : x[i], vm = okm ? xm[i] : x[i];
         g[i] = vp > vm ? (fp - fm) / (vp - vm) : 0;
      }
   }
   free(act);
   return 0;
}

/* H := diag(1 / f''_ii) from one batch of second differences (step 1e-4 (|x_i| + 1): large enough that curvatures down to
 * ~1e-3 rise above the rounding of lnL).  The parameters' curvatures span many orders of magnitude (branch lengths ~1e5,
 * an omega of a small site class ~1e-2), and BFGS started from a multiple of the identity spends dozens of iterations
 * learning that — or stalls on the flat directions.  Variables on a bound, or with no usable curvature, get the median. */
static int diag_inverse_hessian(pamlh *p, const double *x, double f0, const double *lo, const double *hi, double *H, double *xs, double *ls, int *n_eval)
{
   const int n = p->np;
   int i, rc, m = 0, nq = 0;
   double *hd = (double *)malloc(2 * n * sizeof(double)), *srt = hd + n, med;
   int *slot = (int *)malloc(n * sizeof(int));
   for (i = 0; i < n; i++) {
      const double h = 1e-4 * (fabs(x[i]) + 1);
      double *xp, *xm;
      slot[i] = -1;
      if (x[i] + h > hi[i] || x[i] - h < lo[i] || (p->frozen && p->frozen[i])) continue;      /* no room for a symmetric difference / held */
      slot[i] = nq;
      xp = xs + (size_t)(2 * nq) * n; xm = xp + n;
      memcpy(xp, x, n * sizeof(double));
      memcpy(xm, x, n * sizeof(double));
      xp[i] = x[i] + h; xm[i] = x[i] - h;
      nq++;
   }
   if (nq && (rc = batch_eval(p, 2 * nq, xs, ls))) { free(hd); free(slot); return rc; }
   *n_eval += 2 * nq;
   for (i = 0; i < n; i++) {
      hd[i] = 0;
      if (slot[i] >= 0) {
         const double h = 1e-4 * (fabs(x[i]) + 1), lp = ls[2 * slot[i]], lm = ls[2 * slot[i] + 1], c = ((-lp) - 2 * f0 + (-lm)) / (h * h);
         hd[i] = (c > 1e-3 && c < 1e12 && lp > -1e299 && lm > -1e299) ? 1 / c : 0;
      }
      if (hd[i] > 0) srt[m++] = hd[i];
   }
   for (i = 1; i < m; i++) { double v = srt[i]; int j = i - 1; while (j >= 0 && srt[j] > v) { srt[j + 1] = srt[j]; j--; } srt[j + 1] = v; }
   med = m ? srt[m / 2] : 1;
   for (i = 0; i < n * n; i++) H[i] = 0;
   for (i = 0; i < n; i++) H[i * n + i] = hd[i] > 0 ? hd[i] : med;
   free(hd); free(slot);
   return 0;
}

/* Maximise lnL over x (in: start, out: estimate).  Returns 0 when converged, 1 when max_iter was reached, < 0 on error. */
int pamlh_optimize(pamlh *p, double *x, double *lnL, int max_iter, double tol, int verbose, int *n_eval_out)
{
   const int n = p->np, NC = p->opt_lean ? 6 : 12;
   double *lo = (double *)malloc(n * sizeof(double)), *hi = (double *)malloc(n * sizeof(double));
   double *g = (double *)malloc(n * sizeof(double)), *g0 = (double *)malloc(n * sizeof(double)), *d = (double *)malloc(n * sizeof(double));
   double *s = (double *)malloc(n * sizeof(double)), *y = (double *)malloc(n * sizeof(double)), *Hy = (double *)malloc(n * sizeof(double));
   double *H = (double *)malloc((size_t)n * n * sizeof(double));
   double *xs = (double *)malloc((size_t)(2 * n + NC) * n * sizeof(double)), *ls = (double *)malloc((2 * n + NC) * sizeof(double));
   unsigned char *fixed = (unsigned char *)malloc(n);
   double f, fnew = 0, as[16], f_restart = 1e300;
   int it, i, j, k, rc = 0, n_eval = 0, reset = 1, small_steps = 0, status = 1, restarts = 0, fresh = 1, warm = 0;
   if (n == 0) { rc = batch_eval(p, 1, x, lnL); status = 0; goto done; }
   if (pamlh_bounds(p, lo, hi)) { rc = pamlh_fail(p, "internal: bounds do not match np"); goto done; }
   /* A search moves the rate matrices a little at a time — a finite-difference step, a line-search step — and every eigen set is
    * decomposed again and again: the device's Jacobi sweeps start from the set's previous eigenvectors while the search runs
    * (paml_amd_set_eigen_warm_start: 3-5 sweeps instead of 9-10; PAMLH_EIGEN_WARM=0 keeps them cold).  Off again at the end: what
    * follows (the final evaluation's tables, standard errors, BEB) does not depend on the path the search took. */
   if ((rc = pamlh_engine_ready(p))) goto done;
   warm = !(getenv("PAMLH_EIGEN_WARM") && atoi(getenv("PAMLH_EIGEN_WARM")) == 0);
   if (warm) paml_amd_set_eigen_warm_start(p->eng, 1, NULL);
   {
      int start[PAMLH_MAXEIG + 4], len[PAMLH_MAXEIG + 4];
      const int ng = pamlh_simplex_groups(p, start, len, PAMLH_MAXEIG + 4);
      p->opt_transformed = ng > 0;
      if (p->clock || ng) {         /* iterate on (root age, age ratios) and on log-ratios of proportions: see clock_y_to_x */
         double *y = (double *)malloc(n * sizeof(double));
         for (i = 0; i < n; i++) {      /* (not the proportions: their box is that of the transformed variables) */
            int in_group = 0;
            for (j = 0; j < ng; j++) if (i >= start[j] && i < start[j] + len[j]) in_group = 1;
            if (!in_group) x[i] = x[i] < lo[i] ? lo[i] : x[i] > hi[i] ? hi[i] : x[i];
         }
         clock_x_to_y(p, x, y);
         memcpy(x, y, n * sizeof(double));
         free(y);
         if (p->clock) for (i = 0; i < p->ns - 1; i++) { lo[i] = (p->ns + i == p->root) ? (p->tipdate ? p->age_low[p->root] : 0) + 1e-5 : 1e-8; hi[i] = (p->ns + i == p->root) ? 50 : 1; }
         for (j = 0; j < ng; j++) for (i = 0; i < len[j]; i++) { lo[start[j] + i] = -Y_SIMPLEX; hi[start[j] + i] = Y_SIMPLEX; }
      }
   }
   for (i = 0; i < n; i++) x[i] = x[i] < lo[i] ? lo[i] : x[i] > hi[i] ? hi[i] : x[i];
   if ((rc = batch_eval(p, 1, x, ls))) goto done;
   n_eval++;
   f = -ls[0];
   if (f > 1e299) { rc = pamlh_fail(p, "the starting point is infeasible"); goto done; }
   if (verbose) printf("start     lnL %.6f\n", -f);
   if (p->opt_lean == 1) { for (i = 0; i < n * n; i++) H[i] = 0; for (i = 0; i < n; i++) H[i * n + i] = 1; reset = 1; }      /* (2: lean, but from second differences) */
   else {
      if ((rc = diag_inverse_hessian(p, x, f, lo, hi, H, xs, ls, &n_eval))) goto done;
      reset = 0;
   }
   if ((rc = gradient(p, x, f, lo, hi, g, xs, ls, &n_eval))) goto done;
   for (it = 0; it < max_iter; it++) {
      double amax = 1e300, best = f, abest = 0, gd, sy;
      int nc = 0;
      /* variables sitting on a bound with the gradient pushing outward stay there this iteration */
      for (i = 0; i < n; i++) fixed[i] = (x[i] <= lo[i] && g[i] > 0) || (x[i] >= hi[i] && g[i] < 0) || (p->frozen && p->frozen[i]);
      for (i = 0; i < n; i++) {
         d[i] = 0;
         if (fixed[i]) continue;
         for (j = 0; j < n; j++)
            if (!fixed[j]) d[i] -= H[i * n + j] * g[j];
      }
      gd = dot(g, d, n);
      if (!(gd < 0)) {     /* not a descent direction: steepest descent on the free variables */
         for (i = 0; i < n; i++) d[i] = fixed[i] ? 0 : -g[i];
         gd = dot(g, d, n);
         reset = 1;
         if (!(gd < 0)) { status = 0; break; }     /* projected gradient is zero */
      }
      /* the search path is the projection of x + a d on the box: a variable that reaches its bound stays there while the
       * others move on; amax = the step at which the last one is stopped */
      amax = 0;
      for (i = 0; i < n; i++) {
         const double ai = d[i] > 0 ? (hi[i] - x[i]) / d[i] : d[i] < 0 ? (lo[i] - x[i]) / d[i] : 0;
         if (ai > amax) amax = ai;
      }
      /* line search: NC step lengths in one launch; after a reset the scale of d is unknown, so start lower */
      for (k = 0; k < 2 && abest == 0; k++) {
         double a = (reset ? 1.0 / (1 + sqrt(dot(d, d, n))) : 1.0) * 4.0 * (k ? pow(0.5, NC) : 1.0);
         double prev = -1;
         nc = 0;
         for (j = 0; j < NC; j++, a *= 0.5) {
            const double aa = a < amax ? a : amax;
            if (aa == prev) continue;
            prev = aa;
            for (i = 0; i < n; i++) {
               double v = x[i] + aa * d[i];
               xs[(size_t)nc * n + i] = v < lo[i] ? lo[i] : v > hi[i] ? hi[i] : v;
            }
            as[nc] = aa;
            nc++;
         }
         if ((rc = batch_eval(p, nc, xs, ls))) goto done;
         n_eval += nc;
         for (j = 0; j < nc; j++)
            if (-ls[j] < best) { best = -ls[j]; abest = as[j]; }
      }
      if (abest == 0) {         /* no step length improves f */
         if (!fresh && !p->opt_lean) {          /* distrust the curvature information once before giving up */
            if ((rc = diag_inverse_hessian(p, x, f, lo, hi, H, xs, ls, &n_eval))) goto done;
            fresh = 1;
            continue;
         }
         status = 0;
         break;
      }
      fnew = best;
      for (i = 0; i < n; i++) {
         double v = x[i] + abest * d[i];
         v = v < lo[i] ? lo[i] : v > hi[i] ? hi[i] : v;
         s[i] = v - x[i];
         x[i] = v;
      }
      memcpy(g0, g, n * sizeof(double));
      if ((rc = gradient(p, x, fnew, lo, hi, g, xs, ls, &n_eval))) goto done;
      if (verbose) {
         double gf = 0;
         for (i = 0; i < n; i++) if (!((x[i] <= lo[i] && g[i] > 0) || (x[i] >= hi[i] && g[i] < 0))) gf += g[i] * g[i];
         fprintf(stderr, "iter %3d  lnL %.6f  step %.3g  |g free| %.3g  evals %d\n", it + 1, -fnew, abest, sqrt(gf), n_eval);
      }
      /* BFGS update of the inverse Hessian */
      /* (in the subspace of the variables that moved: a variable held on its bound has s = 0, and its gradient change —
       * often the largest of all, e.g. an omega class pinned at 0 — would only corrupt the curvature of the others) */
      for (i = 0; i < n; i++) y[i] = fixed[i] ? 0 : g[i] - g0[i];
      sy = dot(s, y, n);
      if (sy > 1e-14 * sqrt(dot(s, s, n) * dot(y, y, n))) {
         double yHy;
         if (reset) {            /* first update after a reset: scale the identity (Nocedal & Wright 6.20) */
            const double sc = sy / dot(y, y, n);
            for (i = 0; i < n * n; i++) H[i] = 0;
            for (i = 0; i < n; i++) H[i * n + i] = sc;
         }
         for (i = 0; i < n; i++) { Hy[i] = 0; for (j = 0; j < n; j++) Hy[i] += H[i * n + j] * y[j]; }
         yHy = dot(y, Hy, n);
         for (i = 0; i < n; i++)
            for (j = 0; j < n; j++)
               H[i * n + j] += (1 + yHy / sy) * s[i] * s[j] / sy - (Hy[i] * s[j] + s[i] * Hy[j]) / sy;
         reset = 0;
      }
      fresh = 0;
      {
         double smax = 0;
         for (i = 0; i < n; i++) { const double r = fabs(s[i]) / (fabs(x[i]) + 1); if (r > smax) smax = r; }
         small_steps = (f - fnew < tol * (fabs(fnew) + 1) && smax < 1e-5) ? small_steps + 1 : 0;
      }
      if (p->opt_lean && p->opt_abs_tol > 0 && f - fnew < p->opt_abs_tol) { f = fnew; status = 0; break; }
      f = fnew;
      if (small_steps >= 2) {
         /* Two tiny steps in a row: either the maximum, or an inverse Hessian that has gone bad on a ridge (the site-class
          * models have them: proportions against omegas).  Forget the curvature and go on from steepest descent; stop when
          * such a restart (from the diagonal second differences) no longer gains anything. */
         if (!p->opt_lean && restarts < 8 && f_restart - f > 1e-7 * (fabs(f) + 1)) {
            f_restart = f; restarts++; small_steps = 0; fresh = 1;
            if ((rc = diag_inverse_hessian(p, x, f, lo, hi, H, xs, ls, &n_eval))) goto done;
            continue;
         }
         status = 0;
         break;
      }
   }
   *lnL = -f;
   if (p->clock || p->opt_transformed) { double *y = (double *)malloc(n * sizeof(double)); memcpy(y, x, n * sizeof(double)); clock_y_to_x(p, y, x); free(y); }
   p->opt_transformed = 0;
   /* leave the model state at the estimate */
   if (pamlh_set_x(p, x, n)) rc = -1;
done:
   if (warm && p->eng) paml_amd_set_eigen_warm_start(p->eng, 0, NULL);
   p->opt_transformed = 0;
   if (n_eval_out) *n_eval_out = n_eval;
   free(lo); free(hi); free(g); free(g0); free(d); free(s); free(y); free(Hy); free(H); free(xs); free(ls); free(fixed);
   return rc ? rc : status;
}

/* ---- method = 1: one branch at a time ----------------------------------------------------------------------------------------
 * minbranches (treesub.c:8039-8117) restated on paml_amd_eval_branch: cycle through the branches in tree.branches order; for
 * each, Newton's direction p = -l' / |l''| from the branch-local l, l', l'' (lfuntdd), cut to the interval [1e-8, 50], and
 * the step length quartered until lnL improves (lfunt) — here the trial lengths t0 + s p, s = step, step / 4, step / 16,
 * step / 64 are ONE call.  The engine keeps the partials of both sides of every edge resident, so moving on to the next
 * branch recomputes only the nodes between the two (updateconP treesub.c:7982; com.oldconP treespace.c:250).
 * x[0 .. ntime) are updated in place; returns +lnL in *lnL.  e = convergence tolerance of a cycle (e_minbranches). */
int pamlh_minbranches(pamlh *p, double *x, double e, double *lnL, int verbose)
{
   const double tb0 = 1e-8, tb1 = 50, smallv = 1e-20;
   const int maxcycle = 500, ncycleb = 10;
   double *br;
   double L = 0, Lcycle = 0, y[4], dy[4], ddy[4], ts[4];
   int icycle, ib, icb, i, rc = 0, have_L = 0;
   if (p->clock || p->fix_blength || !p->ntime || p->ntime != p->nbranch || p->adg)
      return pamlh_fail(p, "minbranches needs free, unconstrained branch lengths (clock = 0, fix_blength = 0) and no rho");
   if (pamlh_set_x(p, x, p->np) || !pamlh_model_feasible(p)) return pamlh_fail(p, "minbranches: the model rejects x");
   if ((rc = pamlh_engine_model(p))) return rc;
   br = (double *)malloc(p->nnode * sizeof(double));
   memcpy(br, p->branch, p->nnode * sizeof(double));
   for (icycle = 0; icycle < maxcycle; icycle++) {
      for (ib = 0; ib < p->nbranch; ib++) {
         const int b = p->branch_node[ib];
         double t0 = br[b], t = t0, L0 = 0, Lt = 0, d0 = 0, dd0 = 0;
         int have0 = 0;      /* l, l', l'' at t0 are known from the batch of trial lengths that found t0 (every call returns all three) */
         for (icb = 0; icb < ncycleb; icb++) {
            double pn, step, s;
            int found = 0;
            if (!have0) {
               ts[0] = t0;
               if (paml_amd_eval_branch(p->eng, b, 1, ts, br, p->ngene > 1 ? p->rgene : NULL, y, dy, ddy)) { rc = pamlh_fail(p, "%s", paml_amd_last_error(p->eng)); goto done; }
               L0 = y[0]; d0 = dy[0]; dd0 = ddy[0];
            }
            pn = d0 / fabs(dd0);                             /* = -dl / |ddl| with l = -lnL */
            if (!(fabs(pn) >= smallv)) step = 0;             /* (also catches NaN) */
            else if (pn < 0) step = fmin(1, (tb0 - t0) / pn);
            else step = fmin(1, (tb1 - t0) / pn);
            if (icycle == 0 && step != 1 && step != 0) step *= 0.99;      /* keep off the border */
            for (s = step; s > smallv && !found; s /= 256) {
               int k, nt = 0;
               double sk = s;
               for (k = 0; k < 4 && sk > smallv; k++, sk /= 4) ts[nt++] = t0 + sk * pn;
               if (paml_amd_eval_branch(p->eng, b, nt, ts, br, p->ngene > 1 ? p->rgene : NULL, y, dy, ddy)) { rc = pamlh_fail(p, "%s", paml_amd_last_error(p->eng)); goto done; }
               for (k = 0; k < nt; k++)
                  if (y[k] > L0) { t = ts[k]; Lt = y[k]; d0 = dy[k]; dd0 = ddy[k]; found = 1; break; }
            }
            if (!found) { t = t0; Lt = L0; break; }
            if (fabs(t - t0) < e * fabs(1 + t) && fabs(Lt - L0) < e) break;
            t0 = t; L0 = Lt; have0 = 1;
         }
         br[b] = t;
         L = Lt; have_L = 1;
      }
      if (verbose) fprintf(stderr, "\tminbranches cycle %2d: lnL %.6f\n", icycle + 1, L);
      if (icycle && fabs(L - Lcycle) < e) break;
      if (!icycle && !have_L) break;
      Lcycle = L;
   }
   for (i = 0; i < p->ntime; i++) x[i] = br[p->branch_node[i]];
   *lnL = L;
done:
   free(br);
   return rc;
}

/* minB (treesub.c:7826-7941): alternate between the substitution parameters (ming2 on x[ntime..np) with the branch lengths
 * held; here pamlh_optimize with those frozen) and minbranches, tightening the tolerances as the improvement per round falls.
 * Returns 0 converged, 1 round limit, < 0 error.  n_eval: full likelihood evaluations spent by the parameter steps. */
int pamlh_optimize_minb(pamlh *p, double *x, double *lnL, double e0, int verbose, int *n_eval_out)
{
   const int np = p->np, npcom = np - p->ntime, maxr = npcom ? 200 : 1;
   double e = npcom ? 5.0 : e0, e_mb = e, L = 0, L0 = -1e300, dl;
   int ir, i, rc = 0, status = 1, n_eval = 0, ne, calm = 0;
   unsigned char *frozen = (unsigned char *)calloc(np ? np : 1, 1);
   for (i = 0; i < p->ntime; i++) frozen[i] = 1;
   for (ir = 0; ir < maxr; ir++) {
      if (npcom) {
         /* (a round that gained less than the tolerance is checked by one that starts from second differences instead of the
          *  identity: it must have been the maximum, not a badly scaled first step) */
         p->frozen = frozen; p->opt_lean = calm ? 2 : 1; p->opt_abs_tol = e;
         rc = pamlh_optimize(p, x, &L, e > 0.05 ? 2 : 30, 1e-10, 0, &ne);
         p->frozen = NULL; p->opt_lean = 0; p->opt_abs_tol = 0;
         n_eval += ne;
         if (rc < 0) goto done;
         if (verbose) fprintf(stderr, "round %da: parameters, lnL %.6f (%d evaluations)\n", ir + 1, L, ne);
      }
      if ((rc = pamlh_minbranches(p, x, e_mb, &L, verbose > 1))) goto done;
      if (verbose) fprintf(stderr, "round %db: branch lengths, lnL %.6f (e = %.3g)\n", ir + 1, L, e_mb);
      dl = fabs(L - L0);
      calm = (dl < e0 && e <= 0.02) ? calm + 1 : 0;
      if (calm >= 2 || (calm && !npcom)) { status = 0; break; }
      e /= 2; if (dl < 1) e /= 2;
      if (dl < 0.5) e = fmin(e, 1e-3);
      else if (dl > 10) e = fmax(e, 0.1);
      e_mb = fmax(e, 1e-6);
      e = fmax(e, 1e-6);
      L0 = L;
   }
   if (!npcom) status = 0;
   /* "restore things": one ordinary evaluation at the estimate (also leaves the model state there) */
   if (pamlh_set_x(p, x, np)) { rc = -1; goto done; }
   if ((rc = pamlh_eval_gpu(p, lnL, NULL))) goto done;
   n_eval++;
   if (fabs(*lnL - L) > 1e-6 * (fabs(L) + 1)) rc = pamlh_fail(p, "minB: lnL %.9f after the last round, %.9f on re-evaluation", L, *lnL);
done:
   free(frozen);
   p->frozen = NULL; p->opt_lean = 0; p->opt_abs_tol = 0;
   if (n_eval_out) *n_eval_out = n_eval;
   return rc ? rc : status;
}

/* Gauss-Jordan inverse of the nf x nf matrix M (row-major, destroyed); returns 0, or -1 when singular */
static int invert(double *M, int nf, double *inv)
{
   int i, j, k;
   for (i = 0; i < nf; i++)
      for (j = 0; j < nf; j++) inv[i * nf + j] = (i == j);
   for (k = 0; k < nf; k++) {
      int piv = k;
      double d;
      for (i = k + 1; i < nf; i++)
         if (fabs(M[i * nf + k]) > fabs(M[piv * nf + k])) piv = i;
      if (fabs(M[piv * nf + k]) < 1e-300) return -1;
      if (piv != k)
         for (j = 0; j < nf; j++) {
            d = M[k * nf + j]; M[k * nf + j] = M[piv * nf + j]; M[piv * nf + j] = d;
            d = inv[k * nf + j]; inv[k * nf + j] = inv[piv * nf + j]; inv[piv * nf + j] = d;
         }
      d = M[k * nf + k];
      for (j = 0; j < nf; j++) { M[k * nf + j] /= d; inv[k * nf + j] /= d; }
      for (i = 0; i < nf; i++)
         if (i != k && M[i * nf + k] != 0) {
            d = M[i * nf + k];
            for (j = 0; j < nf; j++) { M[i * nf + j] -= d * M[k * nf + j]; inv[i * nf + j] -= d * inv[k * nf + j]; }
         }
   }
   return 0;
}

/* Standard errors at the estimate x (getSE = 1).  hess (may be NULL) receives the information matrix used, np x np.
 *
 * method 0 — what the reference prints: HessianSKT2004 (treesub.c:7241-7307; Seo, Kishino & Thorne 2004), the outer
 *   product of per-pattern scores  I_ij = sum_h w_h d_i(h) d_j(h),  d_i(h) = (log f_h(x + e_i) - log f_h(x - e_i)) / (2 e_i),
 *   e_i = 2 Small_Diff (|x_i| + 1) with Small_Diff = 0.5e-6, inverted as a whole (baseml.c:580-600).  The 2 np perturbed
 *   evaluations, each with its per-pattern log f_h, are ONE batch on the GPU.
 * method 1 — the observed information by central second differences of lnL (Hessian() tools.c:5984), the 2 np^2 + 1 point
 *   stencil again one batch; parameters on a bound are left out and get se = -1. */
int pamlh_standard_errors(pamlh *p, const double *x, int method, double *se, double *hess)
{
   const int n = p->np, npatt = p->npatt;
   double *lo = (double *)malloc((n + 1) * sizeof(double)), *hi = (double *)malloc((n + 1) * sizeof(double)), *h = (double *)malloc((n + 1) * sizeof(double));
   double *H = (double *)calloc((size_t)n * n + 1, sizeof(double)), *A = NULL, *Ai = NULL, *xs = NULL, *ls = NULL, *lf = NULL;
   int *freev = (int *)malloc((n + 1) * sizeof(int)), nf = 0, i, j, rc = 0;
   if (n == 0) goto done;
   if (pamlh_bounds(p, lo, hi)) { rc = pamlh_fail(p, "internal: bounds do not match np"); goto done; }
   for (i = 0; i < n; i++) se[i] = -1;
   if (method == 0 && (!p->fix_rho || p->rho0 != 0)) method = 1;      /* lfunAdG has no per-pattern log f_h: the outer product of scores does not exist */
   if (method == 0) {
      int hp;
      xs = (double *)malloc((size_t)2 * n * n * sizeof(double));
      ls = (double *)malloc((size_t)2 * n * sizeof(double));
      lf = (double *)malloc((size_t)2 * n * npatt * sizeof(double));
      for (i = 0; i < n; i++) {
         h[i] = 1e-6 * (fabs(x[i]) + 1);
         memcpy(xs + (size_t)(2 * i) * n, x, n * sizeof(double));
         memcpy(xs + (size_t)(2 * i + 1) * n, x, n * sizeof(double));
         xs[(size_t)(2 * i) * n + i] = x[i] - h[i];
         xs[(size_t)(2 * i + 1) * n + i] = x[i] + h[i];
      }
      if ((rc = eval_batch_lnf(p, 2 * n, xs, ls, lf))) goto done;
      for (i = 0; i < n; i++)         /* scores, in place over the "minus" rows */
         for (hp = 0; hp < npatt; hp++)
            lf[(size_t)(2 * i) * npatt + hp] = (lf[(size_t)(2 * i + 1) * npatt + hp] - lf[(size_t)(2 * i) * npatt + hp]) / (2 * h[i]);
      for (i = 0; i < n; i++)
         for (j = 0; j <= i; j++) {
            double s = 0;
            for (hp = 0; hp < npatt; hp++) s += lf[(size_t)(2 * i) * npatt + hp] * lf[(size_t)(2 * j) * npatt + hp] * p->w[hp];
            H[i * n + j] = H[j * n + i] = s;
         }
      for (i = 0; i < n; i++) freev[nf++] = i;
   }
   else {
      const size_t npts = (size_t)2 * n * n + 1;
      size_t q = 0;
      xs = (double *)malloc(npts * n * sizeof(double));
      ls = (double *)malloc(npts * sizeof(double));
      for (i = 0; i < n; i++) {
         h[i] = 1e-4 * (fabs(x[i]) + 1);
         if (x[i] - h[i] >= lo[i] && x[i] + h[i] <= hi[i]) freev[nf++] = i;
      }
#define PT(di, si, dj, sj)                                                                                       \
   do {                                                                                                          \
      double *v = xs + q * n;                                                                                    \
      memcpy(v, x, n * sizeof(double));                                                                          \
      v[di] += (si) * h[di];                                                                                     \
      if ((dj) >= 0) v[dj] += (sj) * h[dj];                                                                      \
      q++;                                                                                                       \
   } while (0)
      memcpy(xs, x, n * sizeof(double));
      q = 1;
      for (i = 0; i < nf; i++) { PT(freev[i], +1, -1, 0); PT(freev[i], -1, -1, 0); }
      for (i = 0; i < nf; i++)
         for (j = i + 1; j < nf; j++) { PT(freev[i], +1, freev[j], +1); PT(freev[i], +1, freev[j], -1); PT(freev[i], -1, freev[j], +1); PT(freev[i], -1, freev[j], -1); }
#undef PT
      if ((rc = pamlh_eval_batch_gpu(p, (int)q, xs, ls))) goto done;
      q = 1;
      for (i = 0; i < nf; i++, q += 2) {
         const int a = freev[i];
         H[a * n + a] = -(ls[q] - 2 * ls[0] + ls[q + 1]) / (h[a] * h[a]);
      }
      for (i = 0; i < nf; i++)
         for (j = i + 1; j < nf; j++, q += 4) {
            const int a = freev[i], b = freev[j];
            H[a * n + b] = H[b * n + a] = -(ls[q] - ls[q + 1] - ls[q + 2] + ls[q + 3]) / (4 * h[a] * h[b]);
         }
   }
   if (hess) memcpy(hess, H, (size_t)n * n * sizeof(double));
   A = (double *)malloc((size_t)nf * nf * sizeof(double) + 8);
   Ai = (double *)malloc((size_t)nf * nf * sizeof(double) + 8);
   for (i = 0; i < nf; i++)
      for (j = 0; j < nf; j++) A[i * nf + j] = H[freev[i] * n + freev[j]];
   if (invert(A, nf, Ai)) { rc = pamlh_fail(p, "the information matrix is singular"); goto done; }
   for (i = 0; i < nf; i++) se[freev[i]] = Ai[i * nf + i] > 0 ? sqrt(Ai[i * nf + i]) : -1;
done:
   free(lo); free(hi); free(h); free(xs); free(ls); free(lf); free(H); free(A); free(Ai); free(freev);
   return rc;
}


/* ------------------------------------------------------------------------------------------------------------------------
 * Comparison of the trees of a tree file from their per-pattern log likelihoods (rell treesub.c:5844-6009): the consumer of
 * the `lnf` values of several analyses of the same alignment.
 *   li, Dli = li - l(best), SE of Dli from the sitewise differences, pKH = 1 - Phi(-Dli / SE)   (Kishino & Hasegawa 1989),
 *   pRELL   = share of the bootstrap replicates (sites resampled within each gene, likelihoods re-weighted, not re-estimated) in
 *             which the tree is the best (ties shared),
 *   pSH     = Shimodaira & Hasegawa (1999): replicates centred per tree; share in which max_j l*_j - l*_i exceeds l(best) - l_i.
 * lnf: [n_trees][n_patt]; w: pattern counts; gene_off: n_genes + 1 pattern offsets (NULL: one gene); n_rep: 0 = the reference's
 * choice (10 000 below 10^5 sites).  The replicates come from a generator of this file (xoshiro256**, seeded): pRELL and pSH agree
 * with the reference's to Monte-Carlo accuracy, the other columns digit for digit.  pKH = pSH = -1 for the best tree. */
static unsigned long long tc_s[4];
static unsigned long long tc_next(void)
{
   const unsigned long long r = ((tc_s[1] * 5) << 7 | (tc_s[1] * 5) >> 57) * 9, t = tc_s[1] << 17;
   tc_s[2] ^= tc_s[0]; tc_s[3] ^= tc_s[1]; tc_s[1] ^= tc_s[2]; tc_s[0] ^= tc_s[3]; tc_s[2] ^= t;
   tc_s[3] = tc_s[3] << 45 | tc_s[3] >> 19;
   return r;
}
int pamlh_tree_comparison(int n_trees, int n_patt, const double *w, const double *lnf, int n_genes, const int *gene_off, int n_rep,
                          unsigned long long seed, double *li, double *dli, double *se, double *pkh, double *psh, double *prell, int *best)
{
   int t, h, r, g, k, ml = 0, nbest, one_gene[2];
   long ls = 0;
   double *rep, *mx, y;
   int *sitelist, *cnt, *btrees;
   if (n_trees < 2 || n_patt < 1 || !w || !lnf || n_genes < 1) return -1;
   one_gene[0] = 0; one_gene[1] = n_patt;
   if (!gene_off) { gene_off = one_gene; n_genes = 1; }
   for (h = 0; h < n_patt; h++) ls += (long)w[h];
   if (n_rep <= 0) n_rep = ls < 100000 ? 10000 : 50;
   for (t = 0; t < n_trees; t++) {
      for (h = 0, li[t] = 0; h < n_patt; h++) li[t] += w[h] * lnf[(size_t)t * n_patt + h];
      if (t && li[t] > li[ml]) ml = t;
   }
   for (t = 0; t < n_trees; t++) {
      const double mdl = (li[t] - li[ml]) / ls;
      dli[t] = li[t] - li[ml];
      for (h = 0, se[t] = 0; h < n_patt; h++) {
         y = lnf[(size_t)t * n_patt + h] - lnf[(size_t)ml * n_patt + h];
         se[t] += w[h] * (y - mdl) * (y - mdl);
      }
      se[t] = t == ml ? 0 : sqrt(se[t]);
      pkh[t] = (t == ml || fabs(se[t]) < 1e-6) ? -1 : 1 - pamlh_cdf_normal(-dli[t] / se[t]);
      prell[t] = psh[t] = 0;
   }
   /* bootstrap: sites drawn with replacement inside every gene */
   {  unsigned long long z = seed + 0x9E3779B97F4A7C15ULL;
      for (k = 0; k < 4; k++) { unsigned long long v = (z += 0x9E3779B97F4A7C15ULL); v = (v ^ v >> 30) * 0xBF58476D1CE4E5B9ULL; v = (v ^ v >> 27) * 0x94D049BB133111EBULL; tc_s[k] = v ^ v >> 31; } }
   rep = (double *)calloc((size_t)n_trees * n_rep, sizeof(double));
   mx = (double *)malloc(n_rep * sizeof(double));
   sitelist = (int *)malloc((size_t)ls * sizeof(int));
   cnt = (int *)malloc(n_patt * sizeof(int));
   btrees = (int *)malloc(n_trees * sizeof(int));
   for (h = 0, k = 0; h < n_patt; h++) for (r = 0; r < (int)w[h]; r++) sitelist[k++] = h;
   for (r = 0; r < n_rep; r++) {
      long s0 = 0;
      memset(cnt, 0, n_patt * sizeof(int));
      for (g = 0; g < n_genes; g++) {
         long lg = 0, j;
         for (h = gene_off[g]; h < gene_off[g + 1]; h++) lg += (long)w[h];
         for (j = 0; j < lg; j++) cnt[sitelist[s0 + (long)((tc_next() >> 11) * (1.0 / 9007199254740992.0) * lg)]]++;
         s0 += lg;
      }
      for (h = 0; h < n_patt; h++)
         if (cnt[h]) for (t = 0; t < n_trees; t++) rep[(size_t)t * n_rep + r] += cnt[h] * lnf[(size_t)t * n_patt + h];
      for (t = 1, nbest = 1, btrees[0] = 0, y = rep[r]; t < n_trees; t++) {
         if (fabs(rep[(size_t)t * n_rep + r] - y) < 1e-5) btrees[nbest++] = t;
         else if (rep[(size_t)t * n_rep + r] > y) { nbest = 1; btrees[0] = t; y = rep[(size_t)t * n_rep + r]; }
      }
      for (t = 0; t < nbest; t++) prell[btrees[t]] += 1.0 / ((double)n_rep * nbest);
   }
   for (t = 0; t < n_trees; t++) {      /* S-H: centre each tree's replicates */
      for (r = 0, y = 0; r < n_rep; r++) y += rep[(size_t)t * n_rep + r];
      for (r = 0, y /= n_rep; r < n_rep; r++) rep[(size_t)t * n_rep + r] -= y;
   }
   for (r = 0; r < n_rep; r++) for (t = 1, mx[r] = rep[r]; t < n_trees; t++) if (rep[(size_t)t * n_rep + r] > mx[r]) mx[r] = rep[(size_t)t * n_rep + r];
   for (t = 0; t < n_trees; t++) {
      for (r = 0; r < n_rep; r++) if (mx[r] - rep[(size_t)t * n_rep + r] > li[ml] - li[t]) psh[t] += 1.0 / n_rep;
      if (t == ml || fabs(se[t]) < 1e-6) psh[t] = -1;
   }
   if (best) *best = ml;
   free(rep); free(mx); free(sitelist); free(cnt); free(btrees);
   return 0;
}
