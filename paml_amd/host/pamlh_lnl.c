/* pamlh_lnl — command-line driver: one likelihood evaluation of a codeml/baseml analysis on the MI355X.
 *   usage: pamlh_lnl <codeml|baseml> <file.ctl> [--optimize] [--ancestral] [--gpus N [--devices a,b,...]] [--tree K] [x0 x1 ...]
 *   (--set "key = value": replaces an option of the control file, e.g. one of the site models of an "NSsites = 0 1 2 7 8" list;
 *    --tree K: the K-th tree of the tree file, 1-based; --all-trees: every tree in turn — the reference's loop, Forestry codeml.c:635 —
 *    each optimised from the control file's initial values, then the comparison table of rell(), treesub.c:5844)
 * Reads the control file, the sequence and tree files it names, and the parameter vector from the command line,
 * else from in.codeml / in.baseml beside the ctl (the reference's "-1 x..." single-evaluation recipe, treesub.c:4057),
 * else the ctl's initial values; evaluates lnL through libpaml_amd.so; prints `lnL = ...` like the reference and
 * writes the per-pattern `lnf` file in the reference's layout.  With --optimize the vector is the starting point of a
 * maximum-likelihood search (pamlh_optimize: BFGS with batched finite differences) and the estimates are printed. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/paml_amd.h"
#include "../../include/pamlh.h"

/* --gpus N: one process per GPU.  The parent becomes rank 0; it forks ranks 1 .. N-1 BEFORE anything touches the GPU, obtains the
 * RCCL id and hands it to them through pipes.  Every rank reads the same files, keeps its block of site patterns
 * (pamlh_set_shard) and runs the same code: the all-reduced lnL has the same bits on every rank, so the ranks stay in step
 * without further communication.  Only rank 0 prints.  --devices a,b,...: the GPU of each rank (default: rank r on GPU r).
 * What can go wrong is checked where every rank sees the same answer (more ranks than GPUs: before the fork, by a probe process,
 * since the parent must not initialise HIP before forking; more ranks than reduction chunks: paml_amd_shard_bounds refuses for every
 * rank alike).  A rank that dies on its own (its GPU fails) takes the job with it: rank 0 ends the others and exits non-zero
 * instead of waiting in a collective call for ever. */
#define MAX_RANKS 64
static pid_t child_pid[MAX_RANKS];
static int n_children = 0, ranks_done = 0;

static void end_children(void)      /* atexit of rank 0: an error path left ranks behind */
{
   int i, st;
   if (ranks_done) return;
   for (i = 0; i < n_children; i++) if (child_pid[i] > 0) kill(child_pid[i], SIGTERM);
   for (i = 0; i < n_children; i++) if (child_pid[i] > 0) waitpid(child_pid[i], &st, 0);
   n_children = 0;
}

static void on_sigchld(int sig)     /* a rank ended: fine when it is over, fatal while rank 0 is still working */
{
   int st, i;
   pid_t pid;
   (void)sig;
   while ((pid = waitpid(-1, &st, WNOHANG)) > 0) {
      for (i = 0; i < n_children; i++) if (child_pid[i] == pid) child_pid[i] = 0;
      if (!ranks_done && (!WIFEXITED(st) || WEXITSTATUS(st))) {
         static const char msg[] = "error: a rank ended abnormally; stopping the others\n";
         if (write(2, msg, sizeof(msg) - 1) < 0) {}
         for (i = 0; i < n_children; i++) if (child_pid[i] > 0) kill(child_pid[i], SIGTERM);
         _exit(1);
      }
   }
}

static int probe_device_count(void)      /* in a short-lived process: the parent stays clear of HIP until it has forked its ranks */
{
   int st = 0;
   const pid_t pid = fork();
   if (pid < 0) return -1;
   if (pid == 0) { int n = paml_amd_device_count(); _exit(n > 250 ? 250 : n); }
   if (waitpid(pid, &st, 0) != pid || !WIFEXITED(st)) return -1;
   return WEXITSTATUS(st);
}

static int spawn_ranks(int world, const int *device, int *rank_out, unsigned char *id)
{
   int r, fds[MAX_RANKS][2];
   const int ndev = probe_device_count();
   pid_t pid;
   *rank_out = 0;
   if (world > MAX_RANKS) { fprintf(stderr, "error: --gpus %d: at most %d ranks\n", world, MAX_RANKS); return -1; }
   for (r = 0; r < world; r++)
      if (device[r] < 0 || device[r] >= ndev) { fprintf(stderr, "error: rank %d wants GPU %d but %d are visible\n", r, device[r], ndev); return -1; }
   for (r = 1; r < world; r++) {
      if (pipe(fds[r])) return -1;
      pid = fork();
      if (pid < 0) return -1;
      if (pid == 0) {            /* child = rank r: wait for the id */
         int k;
         n_children = 0;
         for (k = 1; k <= r; k++) close(fds[k][1]);
         *rank_out = r;
         if (paml_amd_set_device(device[r])) { fprintf(stderr, "rank %d: no GPU %d\n", r, device[r]); _exit(1); }
         if (read(fds[r][0], id, PAML_AMD_COMM_ID_BYTES) != PAML_AMD_COMM_ID_BYTES) _exit(1);
         close(fds[r][0]);
         if (!freopen("/dev/null", "w", stdout)) _exit(1);
         return 0;
      }
      child_pid[n_children++] = pid;
      close(fds[r][0]);
   }
   atexit(end_children);
   signal(SIGCHLD, on_sigchld);
   if (paml_amd_set_device(device[0])) return -1;
   if (paml_amd_comm_unique_id(id)) return -1;
   for (r = 1; r < world; r++) {
      if (write(fds[r][1], id, PAML_AMD_COMM_ID_BYTES) != PAML_AMD_COMM_ID_BYTES) return -1;
      close(fds[r][1]);
   }
   return 0;
}

int main(int argc, char **argv)
{
   pamlh *p;
   char err[512];
   double x[4096], lnL, *lnf;
   int np, ntime, npatt, i, nx = 0, optimize = 0, ancestral = 0, gpus = 0, rank = 0, itree = 0, all_trees = 0;
   char over[2048] = "";
   unsigned char comm_id[PAML_AMD_COMM_ID_BYTES];
   int device[MAX_RANKS];
   for (i = 0; i < MAX_RANKS; i++) device[i] = i;
   if (argc < 3) { fprintf(stderr, "usage: %s <codeml|baseml> <ctl> [--optimize] [--ancestral] [--gpus N] [--tree K | --all-trees] [--set 'key = value'] [x...]\n", argv[0]); return 2; }
   for (i = 3; i < argc && nx < 4096; i++) {
      if (!strcmp(argv[i], "--optimize")) optimize = 1;
      else if (!strcmp(argv[i], "--ancestral")) ancestral = 1;
      else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
      else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {      /* the GPU of each rank, e.g. "4,5,6,7" */
         int k = 0;
         char *tok = strtok(argv[++i], ",");
         for (; tok && k < MAX_RANKS; tok = strtok(NULL, ",")) device[k++] = atoi(tok);
      }
      else if (!strcmp(argv[i], "--tree") && i + 1 < argc) itree = atoi(argv[++i]) - 1;
      else if (!strcmp(argv[i], "--all-trees")) all_trees = 1;
      else if (!strcmp(argv[i], "--set") && i + 1 < argc) {      /* --set "NSsites = 2": replaces the control file's option */
         if (strlen(over) + strlen(argv[i + 1]) + 2 >= sizeof(over)) { fprintf(stderr, "error: too many --set options\n"); return 2; }
         strcat(over, argv[++i]); strcat(over, "\n");
      }
      else x[nx++] = atof(argv[i]);
   }
   if (all_trees) {      /* every tree of the file: maximum likelihood on each, then the comparison of rell() from the per-pattern values */
      int nt = 1, t, npt = 0, ng = 1;
      double *all = NULL, *w = NULL, *res;
      const int *goff = NULL;
      for (t = 0; t < nt; t++) {
         int n_eval = 0;
         if (pamlh_load_with(&p, argv[2], argv[1], t, over, err, sizeof(err))) { fprintf(stderr, "error: %s\n", err); return 1; }
         pamlh_dims(p, NULL, NULL, &npatt, NULL, NULL, NULL, NULL, NULL, &np, &ntime);
         if (t == 0) {
            nt = pamlh_n_trees(p); npt = npatt;
            all = (double *)malloc((size_t)nt * npt * sizeof(double)); w = (double *)malloc(npt * sizeof(double));
            memcpy(w, pamlh_weights(p), npt * sizeof(double));
            ng = pamlh_genes(p, &goff, NULL, NULL, NULL);
            if (ng > 1) { int *cp = (int *)malloc((ng + 1) * sizeof(int)); memcpy(cp, goff, (ng + 1) * sizeof(int)); goff = cp; } else goff = NULL;
         }
         pamlh_default_x(p, x, 4096);
         if (np > 0 && pamlh_optimize(p, x, &lnL, 500, 1e-10, 0, &n_eval) < 0) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
         if (pamlh_set_x(p, x, np) || pamlh_eval_gpu(p, &lnL, all + (size_t)t * npt)) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
         printf("TREE # %2d:  lnL(ntime:%3d  np:%3d): %13.6f\n", t + 1, ntime, np, lnL);
         for (i = 0; i < np; i++) printf(" %.6f", x[i]);
         printf("\n");
         pamlh_free(p);
      }
      if (nt > 1) {
         int best = 0;
         res = (double *)malloc((size_t)6 * nt * sizeof(double));
         if (pamlh_tree_comparison(nt, npt, w, all, ng, goff, 0, 20260927ULL, res, res + nt, res + 2 * nt, res + 3 * nt, res + 4 * nt, res + 5 * nt, &best)) return 1;
         printf("\nTree comparisons (Kishino & Hasegawa 1989; Shimodaira & Hasegawa 1999)\n\n%6s %12s %9s %9s%8s%10s%9s\n\n", "tree", "li", "Dli", " +- SE", "pKH", "pSH", "pRELL");
         for (t = 0; t < nt; t++)
            printf("%6d%c%12.3f %9.3f %9.3f%8.3f%10.3f%9.3f\n", t + 1, t == best ? '*' : ' ', res[t], res[nt + t], res[2 * nt + t], res[3 * nt + t], res[4 * nt + t], res[5 * nt + t]);
         free(res);
      }
      free(all); free(w);
      return 0;
   }
   if (gpus > 0 && spawn_ranks(gpus, device, &rank, comm_id)) { fprintf(stderr, "error: could not start %d ranks (GPUs visible: %d; librccl.so.1 present?)\n", gpus, paml_amd_device_count()); return 1; }
   if (pamlh_load_with(&p, argv[2], argv[1], itree, over, err, sizeof(err))) { fprintf(stderr, "error: %s\n", err); return 1; }
   if (gpus > 0 && pamlh_set_shard(p, rank, gpus, comm_id)) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
   pamlh_dims(p, NULL, NULL, &npatt, NULL, NULL, NULL, NULL, NULL, &np, &ntime);
   if (gpus > 0 && (ancestral || pamlh_mgene(p) == 1)) { fprintf(stderr, "error: --gpus gives lnL and estimates; per-site outputs and Mgene = 1 need the whole alignment on one GPU\n"); return 1; }
   if (pamlh_mgene(p) == 1) {      /* separate analyses: every gene on its own (start values: the control file's), lnL summed */
      const int ng = pamlh_genes(p, NULL, NULL, NULL, NULL);
      double sum = 0;
      int g;
      for (g = 0; g < ng; g++) {
         pamlh *q;
         int npg, n_eval = 0, lsg, npattg;
         if (pamlh_gene_subset(p, g, &q)) { fprintf(stderr, "error: gene %d\n", g + 1); return 1; }
         pamlh_dims(q, NULL, NULL, &npattg, NULL, NULL, NULL, NULL, &lsg, &npg, NULL);
         pamlh_default_x(q, x, 4096);
         if (optimize ? pamlh_optimize(q, x, &lnL, 500, 1e-10, 0, &n_eval) < 0 : (pamlh_set_x(q, x, npg) || pamlh_eval_gpu(q, &lnL, NULL))) {
            fprintf(stderr, "error: %s\n", pamlh_error(q)); return 1;
         }
         printf("Gene %2d  ls:%5d  npatt:%4d  lnL = %.6f\nx:", g + 1, lsg, npattg, lnL);
         for (i = 0; i < npg; i++) printf(" %.6f", x[i]);
         printf("\n");
         sum += lnL;
         pamlh_free(q);
      }
      printf("Sum of lnL over the %d genes = %.6f\n", ng, sum);
      pamlh_free(p);
      return 0;
   }
   if (!nx) nx = pamlh_read_inx(p, x, 4096);
   if (!nx) nx = pamlh_default_x(p, x, 4096);
   if (nx != np) { fprintf(stderr, "error: the model has %d parameters (ntime %d) but %d values were given\n", np, ntime, nx); return 1; }
   if (optimize) {
      /* method = 1 in the control file: minB / minbranches (one branch at a time on the branch-local derivatives) */
      const int method1 = pamlh_method(p) == 1;
      int n_eval = 0, rc = method1 ? pamlh_optimize_minb(p, x, &lnL, 1e-6, 1, &n_eval) : pamlh_optimize(p, x, &lnL, 500, 1e-10, 1, &n_eval);
      if (rc < 0) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
      printf("%s after %d likelihood evaluations\nx:", rc ? "iteration limit reached" : "converged", n_eval);
      for (i = 0; i < np; i++) printf(" %.6f", x[i]);
      printf("\n");
      for (i = ntime; i < np; i++) { char nm[64]; pamlh_param_name(p, i, nm, sizeof(nm)); printf("  %-24s %12.6f\n", nm, x[i]); }
      {
         int nnode = 0;
         char *nw;
         pamlh_dims(p, NULL, NULL, NULL, &nnode, NULL, NULL, NULL, NULL, NULL, NULL);
         nw = (char *)malloc((size_t)160 * nnode + 256);
         if (!pamlh_set_x(p, x, np) && !pamlh_newick(p, nw, 160 * nnode + 256)) printf("%s\n", nw);
         free(nw);
      }
      if (!gpus) {      /* (the scores' outer product is a sum over all site patterns: one GPU holds them all) */
         double *se = (double *)malloc((np + 1) * sizeof(double));
         if (!pamlh_standard_errors(p, x, 0, se, NULL)) {
            printf("SEs for parameters:\n ");
            for (i = 0; i < np; i++) printf(" %.6f", se[i]);
            printf("\n");
         }
         free(se);
      }
   }
   if (pamlh_set_x(p, x, np)) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
   lnf = (double *)malloc(npatt * sizeof(double));
   if (pamlh_eval_gpu(p, &lnL, lnf)) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
   printf("ntime & np: %d %d   npatt %d\nlnL  = %.6f\n", ntime, np, npatt, lnL);
   if (gpus > 0) {      /* sharded run: lnL (the total over the ranks) and the estimates are the output */
      int st = 0, bad = 0;
      printf("(site patterns sharded over %d GPUs, %d on rank 0; lnL is the all-reduced total)\n", gpus, npatt);
      fflush(stdout);
      free(lnf);
      pamlh_free(p);
      if (rank == 0) {      /* the result is out: from here on the ranks just end (on_sigchld reaps them too; ECHILD ends the loop) */
         ranks_done = 1;
         signal(SIGCHLD, SIG_DFL);
         while (wait(&st) > 0) bad |= !WIFEXITED(st) || WEXITSTATUS(st);
      }
      return bad ? 1 : 0;
   }
   {  /* codon models without site classes: the reference's "dN & dS for each branch" table (codeml.c:1361-1404) */
      int nn = 0, b;
      double *tab;
      pamlh_dims(p, NULL, NULL, NULL, &nn, NULL, NULL, NULL, NULL, NULL, NULL);
      tab = (double *)malloc((size_t)nn * 6 * sizeof(double));
      if (!strcmp(argv[1], "codeml") && !pamlh_dnds(p, x, tab)) {
         double dnt = 0, dst = 0;
         printf("\ndN & dS for each branch\n\n%7s%11s%8s%8s%8s%8s%8s\n\n", "branch", "t", "N", "S", "dN/dS", "dN", "dS");
         for (b = 0; b < nn - 1; b++) {
            const double *r = tab + b * 6;
            printf("%7d %10.3f %7.1f %7.1f %7.4f %7.4f %7.4f\n", b + 1, r[0], r[1], r[2], r[3], r[4], r[5]);
            dnt += r[4]; dst += r[5];
         }
         printf("\ntree length for dN: %12.4f\ntree length for dS: %12.4f\n", dnt, dst);
      }
      free(tab);
      if (pamlh_set_x(p, x, np)) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
   }
   {  /* site-class models: the NEB table the reference prints (sites with Pr(last class) > 0.5 when its omega > 1) */
      int mode, K, n_sites, h, npos = pamlh_positive_classes(p);
      pamlh_model(p, &mode, &K, NULL, NULL);
      if (mode == 1 && K > 1 && npos && pamlh_class_omega(p)[K - 1] > 1) {
         double *post = (double *)malloc((size_t)K * npatt * sizeof(double)), *mw = (double *)malloc(npatt * sizeof(double));
         const int *pose = pamlh_pose(p, &n_sites);
         if (!pamlh_neb(p, post, mw)) {
            printf("\nNaive Empirical Bayes (NEB): sites with Pr(w>1) > 0.5\n   site   Pr(w>1)   post mean w\n");
            for (h = 0; h < n_sites; h++) {      /* branch-site models: classes 2a + 2b (foreground omega) */
               const double pr = post[(size_t)(K - 1) * npatt + pose[h]] + (npos == 2 ? post[(size_t)(K - 2) * npatt + pose[h]] : 0);
               if (pr > 0.5) printf("%7d   %7.3f   %9.3f\n", h + 1, pr, mw[pose[h]]);
            }
         }
         if (npos == 2) {  /* branch-site model A: BEB over (p0, p1, w0, w2), classes 2a + 2b */
            double *bp = (double *)malloc((size_t)4 * npatt * sizeof(double));
            if (!pamlh_beb_acd(p, x, bp)) {
               printf("\nBayes Empirical Bayes (BEB): positive sites for foreground lineages Prob(w>1) > 0.5\n   site   Pr(w>1)\n");
               for (h = 0; h < n_sites; h++) {
                  const double pr = bp[(size_t)2 * npatt + pose[h]] + bp[(size_t)3 * npatt + pose[h]];
                  if (pr > 0.5) printf("%7d   %7.3f%s\n", h + 1, pr, pr > 0.99 ? "**" : pr > 0.95 ? "*" : "");
               }
            }
            free(bp);
         }
         if (npos == 1) {  /* M2a / M8: the BEB table as well */
            double *pr = (double *)malloc(npatt * sizeof(double)), *sw = (double *)malloc(npatt * sizeof(double));
            if (!pamlh_beb(p, x, pr, mw, sw)) {
               printf("\nBayes Empirical Bayes (BEB): sites with Pr(w>1) > 0.5\n   site   Pr(w>1)   post mean +- SE for w\n");
               for (h = 0; h < n_sites; h++)
                  if (pr[pose[h]] > 0.5) printf("%7d   %7.3f   %9.3f +- %5.3f\n", h + 1, pr[pose[h]], mw[pose[h]], sw[pose[h]]);
            }
            free(pr); free(sw);
         }
         free(post); free(mw);
      }
   }
   if (ancestral) {   /* RateAncestor = 1: most probable state and its probability at every internal node, per site (rst's marginal table) */
      static const char *const alpha[3] = {"TCAG", "", "ARNDCQEGHILKMFPSTWYV"};
      int n, ns, nnode, seqtype_n, n_sites, h, node, k;
      const int *pose = pamlh_pose(p, &n_sites);
      double *post, *best_p;
      int *best;
      pamlh_dims(p, &n, &ns, NULL, &nnode, NULL, NULL, NULL, NULL, NULL, NULL);
      seqtype_n = n == 4 ? 0 : n == 20 ? 2 : 1;
      post = (double *)malloc((size_t)npatt * n * sizeof(double));
      best = (int *)malloc((size_t)(nnode - ns) * npatt * sizeof(int));
      best_p = (double *)malloc((size_t)(nnode - ns) * npatt * sizeof(double));
      for (node = ns; node < nnode; node++) {
         if (pamlh_node_posterior(p, node, post)) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
         for (h = 0; h < npatt; h++) {
            int bi = 0;
            for (k = 1; k < n; k++) if (post[(size_t)h * n + k] > post[(size_t)h * n + bi]) bi = k;
            best[(size_t)(node - ns) * npatt + h] = bi; best_p[(size_t)(node - ns) * npatt + h] = post[(size_t)h * n + bi];
         }
      }
      printf("\nMarginal reconstruction of ancestral states: site, then for nodes %d..%d the most probable state (probability)\n", ns + 1, nnode);
      for (h = 0; h < n_sites; h++) {
         printf("%6d ", h + 1);
         for (node = ns; node < nnode; node++) {
            const int b = best[(size_t)(node - ns) * npatt + pose[h]];
            if (seqtype_n == 1) printf(" %2d(%.3f)", b, best_p[(size_t)(node - ns) * npatt + pose[h]]);      /* codons: index among the sense codons */
            else printf(" %c(%.3f)", alpha[seqtype_n][b], best_p[(size_t)(node - ns) * npatt + pose[h]]);
         }
         printf("\n");
      }
      free(post); free(best); free(best_p);
   }
   pamlh_write_lnf(p, "lnf", lnf);
   free(lnf);
   pamlh_free(p);
   return 0;
}
