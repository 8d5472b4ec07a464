/* pamlh.h — host side of the MI355X likelihood engine, in C as the reference is.
 *
 * It reads the same inputs the reference programs read — the `.ctl` file (GetOptions codeml.c:1694 / baseml.c:954),
 * the sequence file (ReadSeq treesub.c:487: PHYLIP sequential / interleaved, `.` repeats, the `P` pattern format, options G / GC;
 * aligned FASTA and NEXUS, GetSeqFileType treesub.c:367),
 * the tree file (ReadTreeN treesub.c:3048) and `in.codeml` / `in.baseml` (readx treesub.c:4035) — compresses sites
 * into patterns in the reference's order (PatternWeight treesub.c:1386), encodes them (EncodeSeqs 1116,
 * SetMapAmbiguity 1218), and turns a parameter vector x[] into the engine's inputs the way SetParameters
 * (codeml.c:2757, baseml.c:1306) does: branch lengths, pi, eigen systems, site classes.  It then drives
 * libpaml_amd.so through include/paml_amd.h.  Scope (this round): codeml seqtype 1 (icode 0 and 1; CodonFreq 0-5 (incl. F1x4MG / F3x4MG); NSsites
 * 0-13 and 22 with model 0; with '#' labels in the tree the branch model (model 2, NSsites 0), the branch-site models A and B
 * (model 2, NSsites 2 / 3) and the clade models C and D (model 3, NSsites 2 / 3)) and seqtype 2 / 3 (aa models 0,1,2,3), baseml models JC69,K80,F81,F84,HKY85,T92,TN93,REV,UNREST; +Gamma, auto-discrete-gamma (rho), nhomo 1 (base frequencies as parameters); several genes
 * (option G / GC of the sequence file) with Mgene 0,1,2,3,4 for baseml, codeml M0 and aaml;
 * clock 0 and 1 (global clock: x holds the internal node ages); fix_blength 0, 2 (fixed) and 3 (proportional); cleandata 0/1; sequential and interleaved (I) PHYLIP, the P pattern format.  Anything else fails with a message
 * instead of guessing.
 */
#ifndef PAMLH_H
#define PAMLH_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pamlh pamlh;

/* program: "codeml" or "baseml".  Paths inside the ctl are resolved relative to the ctl file's directory. */
int pamlh_load(pamlh **out, const char *ctl_path, const char *program, char *err, int errcap);
/* the same with the tree_index-th (0-based) tree of the tree file — the reference evaluates the trees of the file one after the
 * other (Forestry codeml.c:635, baseml.c:451) — and the number of trees the file holds */
int pamlh_load_tree(pamlh **out, const char *ctl_path, const char *program, int tree_index, char *err, int errcap);
int pamlh_n_trees(const pamlh *p);
/* ... and with control-file options replaced or added: overrides = "key = value" lines separated by newlines or ';' (NULL: none).  A
 * control file that lists several site models ("NSsites = 0 1 2 7 8": the reference runs them in turn, the insmodel loop codeml.c:657-906) is one analysis
 * per model here: overrides = "NSsites = 2".  pamlh_ctl_option returns an option's text as the control file (or an override) gave it. */
int pamlh_load_with(pamlh **out, const char *ctl_path, const char *program, int tree_index, const char *overrides, char *err, int errcap);
const char *pamlh_ctl_option(const pamlh *p, const char *key);
/* "dN & dS for each branch" (DetailOutput codeml.c:1349-1404) under the codon models without site classes: out[n_branches][6] = t, N, S,
 * omega, dN, dS at the parameter vector x; n_branches = n_nodes - 1, in the order of the branch lengths in x (first appearance in the tree file) */
int pamlh_dnds(pamlh *p, const double *x, double *out);
/* Comparison of the trees of a tree file from the per-pattern log likelihoods of their analyses (rell treesub.c:5844-6009: the table
 * "tree li Dli +- SE pKH pSH pRELL" of a run with several trees).  lnf [n_trees][n_patt], w pattern counts, gene_off n_genes + 1 offsets
 * (NULL: one gene); n_rep = 0: the reference's number of bootstrap replicates; outputs have n_trees entries each (pKH, pSH = -1 for the best
 * tree, whose index goes to *best).  li, Dli, SE, pKH are deterministic; pSH and pRELL come from seeded resampling. */
int pamlh_tree_comparison(int n_trees, int n_patt, const double *w, const double *lnf, int n_genes, const int *gene_off, int n_rep,
                          unsigned long long seed, double *li, double *dli, double *se, double *pkh, double *psh, double *prell, int *best);
void pamlh_free(pamlh *p);
const char *pamlh_error(const pamlh *p);

/* sizes: n_states (com.ncode), n_tips (com.ns), n_patt (com.npatt), n_nodes (tree.nnode), root, n_codes, cleandata,
 * ls (sites or codons after cleaning), np (free parameters the reference would put in x[]), ntime (branch lengths in x) */
int pamlh_dims(const pamlh *p, int *n_states, int *n_tips, int *n_patt, int *n_nodes, int *root, int *n_codes,
               int *cleandata, int *ls, int *np, int *ntime);

/* data (valid until pamlh_free) */
const unsigned char *pamlh_tips(const pamlh *p);        /* [n_tips][n_patt] */
const double *pamlh_weights(const pamlh *p);            /* [n_patt] */
const int *pamlh_n_chara(const pamlh *p);               /* [n_codes] */
const unsigned char *pamlh_chara_map(const pamlh *p);   /* [n_codes][n_states] */
const int *pamlh_sons_ptr(const pamlh *p);              /* [n_nodes+1] */
const int *pamlh_sons(const pamlh *p);
const int *pamlh_labels(const pamlh *p);                /* [n_nodes] */
const unsigned char *pamlh_scale_nodes(const pamlh *p); /* [n_nodes], SetNodeScale treesub.c:7177 */
const int *pamlh_branch_order(const pamlh *p);          /* [ntime_all]: node below the i-th branch of tree.branches */

/* default x[]: branch lengths from the tree file, substitution parameters from the ctl (GetInitials) */
int pamlh_default_x(const pamlh *p, double *x, int cap);
/* in.codeml / in.baseml next to the ctl ("-1 x0 x1 ..."); returns the number of values read, 0 if absent */
int pamlh_read_inx(const pamlh *p, double *x, int cap);

/* SetParameters: x -> model state */
int pamlh_set_x(pamlh *p, const double *x, int np);
int pamlh_model(const pamlh *p, int *mode, int *K, int *n_eigen, int *n_labels);
const double *pamlh_branch(const pamlh *p);             /* [n_nodes] */
const double *pamlh_pi(const pamlh *p);                 /* [n_states] */
const double *pamlh_freqK(const pamlh *p);
const double *pamlh_rate(const pamlh *p);
const int *pamlh_eigen_of(const pamlh *p);              /* [K][n_labels] */
const double *pamlh_qfactor(const pamlh *p);
const double *pamlh_adg_matrix(const pamlh *p);         /* [K][K] auto-discrete-gamma transition matrix (rho != 0), else NULL */
/* option G (several genes): returns n_genes; gene_off[n_genes + 1] = first pattern of each gene (com.posG), gene_rate[n_genes]
 * = com.rgene after pamlh_set_x, n_pi = frequency vectors in pamlh_pi (1, or n_genes under Mgene 2 / 4), gene_eigen_of
 * [n_genes][K] = eigen system of (gene, class) (NULL with one gene).  Any output pointer may be NULL. */
/* Mgene = 1 (separate analyses): gene g of the data set as an independent analysis (own patterns, frequencies, parameters); the
 * caller frees it with pamlh_free. */
int pamlh_gene_subset(const pamlh *p, int g, pamlh **out);
int pamlh_mgene(const pamlh *p);                       /* the Mgene option in effect (0 with one gene) */
int pamlh_malpha(const pamlh *p);      /* 1: a gamma shape per gene (Malpha); pamlh_rate() then returns [n_genes][K] */
int pamlh_genes(const pamlh *p, const int **gene_off, const double **gene_rate, int *n_pi, const int **gene_eigen_of);            /* [K][n_labels] time scale per (class, branch type); NULL = all 1 */
/* eigen system i: kind (paml_amd.h), and pointers (NULL when not applicable) */
int pamlh_eigen(const pamlh *p, int i, int *kind, int *nR, double *kappa, const double **U, const double **V,
                const double **Root, const double **Cijk);

/* Multi-GPU, one process per GPU: keep this rank's contiguous block of site patterns and join the ranks' RCCL communicator when the
 * engine is created (id128 = rank 0's paml_amd_comm_unique_id, handed over by any means); every evaluation then returns the
 * lnL of the WHOLE alignment on every rank, bit for bit the same for any number of ranks.  `pamlh_lnl --gpus N` is the driver. */
int pamlh_set_shard(pamlh *p, int rank, int world, const void *id128);

/* The paml_amd_engine behind this analysis (NULL before the first evaluation). */
void *pamlh_engine_handle(const pamlh *p);

/* One likelihood evaluation on the GPU through the engine ABI (creates the engine on first use).  lnf may be NULL. */
int pamlh_eval_gpu(pamlh *p, double *lnL, double *lnf);
/* lnL at n_batch parameter vectors xs[n_batch][np] in ONE launch (paml_amd_eval_batch): vectors that differ only in branch
 * lengths share one model set-up; vectors the model rejects get -1e300. */
int pamlh_eval_batch_gpu(pamlh *p, int n_batch, const double *xs, double *lnL);
/* Box constraints of x[] as SetxBound sets them (codeml.c:1880, baseml.c:1100). */
int pamlh_bounds(const pamlh *p, double *lo, double *hi);
/* Maximum-likelihood estimation (the job of ming2, tools.c:6595): BFGS in the box, every gradient (2 np central
 * differences) and every line search (12 trial steps) evaluated as one batch on the GPU.  x: start in, estimate out.
 * Returns 0 converged, 1 max_iter reached, < 0 error.  n_eval (may be NULL): likelihood evaluations spent. */
int pamlh_optimize(pamlh *p, double *x, double *lnL, int max_iter, double tol, int verbose, int *n_eval);

/* method = 1 of the control file: minB / minbranches (treesub.c:7826, 8039) — the branch lengths are optimised one at a time
 * by Newton steps on the branch-local lnL, dlnL/dt, d2lnL/dt2 (paml_amd_eval_branch; the engine keeps the partials of both
 * sides of every edge resident, so a step along the tree costs the nodes on the path, not the tree), alternating with BFGS
 * on the other parameters.  pamlh_minbranches: one such pass over the branch lengths in x (tolerance e), the rest of x held.
 * pamlh_optimize_minb: the alternation until |delta lnL| < e0.  n_eval: full evaluations spent on the other parameters. */
int pamlh_method(const pamlh *p);      /* the control file's `method` */
int pamlh_minbranches(pamlh *p, double *x, double e, double *lnL, int verbose);
int pamlh_optimize_minb(pamlh *p, double *x, double *lnL, double e0, int verbose, int *n_eval);

/* Standard errors of the estimates (getSE = 1).  method 0: HessianSKT2004 (treesub.c:7241), the outer product of per-pattern
 * scores that the reference's programs print; method 1: observed information from central second differences of lnL
 * (Hessian() tools.c:5984).  Either way all perturbed evaluations are one batch.  se[np] (-1: not available, e.g. a parameter
 * on its bound under method 1); hess: NULL or np x np, the information matrix that was inverted. */
int pamlh_standard_errors(pamlh *p, const double *x, int method, double *se, double *hess);

/* Naive empirical Bayes posterior probabilities of the site classes (lfunNSsites_rate codeml.c:5241) at the model state of
 * the last pamlh_set_x, from the device's fhK: post[K][n_patt]; mean_w[n_patt] (may be NULL) = posterior mean omega.
 * pamlh_pose maps a site (after cleaning) to its pattern; pamlh_class_omega gives the omega of every class. */
int pamlh_neb(pamlh *p, double *post, double *mean_w);
/* Bayes empirical Bayes for M2a / M8 at the estimates x (lfunNSsites_M2M8 codeml.c:6387): posterior probability of the
 * w > 1 class, posterior mean and sd of omega, per pattern [n_patt].  f(x_h | w) for the grid's omegas is one evaluation on
 * the device; the 10^4-point grid sums run on the host. */
/* Reporting: the name of parameter i of x[] ("t 6..7", "kappa", "p0", "w2 (foreground)", ...), and the tree in Newick form with
 * the branch lengths of the current model state (buf needs 64 bytes per node + 128). */
int pamlh_param_name(const pamlh *p, int i, char *buf, int cap);
int pamlh_newick(const pamlh *p, char *buf, int cap);

/* com.plfun's calling convention (codeml.c:125 / baseml.c:70): SetParameters(x) + one likelihood evaluation on the GPU,
 * returns -lnL (what ming2 minimises); +1e300 on error (see pamlh_error). */
double pamlh_plfun(pamlh *p, const double *x, int np);

/* mcmctree's exact-likelihood seam (usedata = 1; lnpD_locus mcmctree.c:1130-1166 -> com.plfun(NULL, -1)): lnL of the locus for
 * node ages age[n_nodes] and either the locus rate rgene (clock = 1) or per-branch rates rate[n_nodes] (clock = 2 / 3), at the
 * substitution parameters of the last pamlh_set_x.  model_changed = 0: only the branch lengths are sent. */
int pamlh_lnpd_locus(pamlh *p, const double *age, double rgene, const double *rate, int model_changed, double *lnL);

/* Marginal ancestral reconstruction (RateAncestor = 1; AncestralMarginal treesub.c:6288) at the current model state:
 * post[n_patt][n_states] = Pr(state at internal node `node` (0-based, >= n_tips) | pattern). */
int pamlh_node_posterior(pamlh *p, int node, double *post);
/* Joint ancestral reconstruction (Pupko et al. 2000; the reference's "(2) Joint reconstruction" in rst): the most probable
 * assignment of states to all internal nodes per pattern — states[n_patt][n_nodes - n_tips] in node order — and its probability. */
int pamlh_joint_reconstruction(pamlh *p, int *states, double *prob);
int pamlh_beb(pamlh *p, const double *x, double *pr_pos, double *mean_w, double *se_w);
/* BEB under branch-site model A and clade models C / D with two branch types (lfunNSsites_ACD codeml.c:6827):
 * post[nc][n_patt] = posterior of the site classes; nc = 4 for A (classes 0, 1, 2a, 2b: Pr(positive selection on the foreground)
 * = post[2] + post[3]), 3 for C and D. */
int pamlh_beb_acd(pamlh *p, const double *x, double *post);
const int *pamlh_pose(const pamlh *p, int *n_sites);
const double *pamlh_class_omega(const pamlh *p);
int pamlh_positive_classes(const pamlh *p);             /* trailing classes that allow omega > 1 (2 for branch-site: 2a + 2b) */

/* Write the reference's `lnf` file layout (print_lnf_site treesub.c:7598) for the last pamlh_eval_gpu. */
int pamlh_write_lnf(const pamlh *p, const char *path, const double *lnf);

#ifdef __cplusplus
}
#endif
#endif
