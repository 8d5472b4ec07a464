/* oracle/cpu_ref.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar IEEE double, reference loop nest and summation order) of the
 * PAML likelihood hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this; the product (libpaml_amd.so and the host driver) never links or calls it.
 *
 * Every function cites the reference file:line (under /root/reference/src) whose arithmetic it
 * restates.  Parity status: PINNED — checked against lnL and per-pattern `lnf` vectors produced in
 * this container by the unmodified reference binaries (oracle/_ref/codeml, oracle/_ref/baseml built
 * by oracle/Makefile); vectors and the generating script live in tests/golden/.
 */
#ifndef PAML_ORACLE_CPU_REF_H
#define PAML_ORACLE_CPU_REF_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_EIGEN_UVROOT = 0, ORC_EIGEN_CIJK = 1, ORC_EIGEN_K80 = 2, ORC_EIGEN_JC69LIKE = 3, ORC_EIGEN_QMAT = 4 };
enum { ORC_MODE_LFUN = 0, ORC_MODE_LFUNDG = 1 };

/* One eigen system = what GetPMatBranch (treesub.c:7503 / 7534) picks for a branch. */
typedef struct {
   int kind;             /* ORC_EIGEN_* */
   int nR;               /* CIJK: number of distinct roots (baseml.c:123 nR) */
   double kappa;         /* K80 */
   const double *U;      /* UVROOT: U[i*n+k]   (tools.c:516);  QMAT: the rate matrix Q[i*n+j] */
   const double *V;      /* UVROOT: V[k*n+j] */
   const double *Root;   /* UVROOT: Root[n];  CIJK: Root[nR] */
   const double *Cijk;   /* CIJK: Cijk[i*n*nR + j*nR + k]  (baseml.c:1572) */
} orc_eigen;

typedef struct {
   int n;                /* states: com.ncode */
   int n_tips;           /* com.ns */
   int n_nodes;          /* tree.nnode; tips are 0..n_tips-1 */
   int root;             /* tree.root */
   int n_patt;           /* com.npatt */
   const int *sons_ptr;  /* CSR over nodes: sons of node i are sons[sons_ptr[i] .. sons_ptr[i+1]) in nodes[i].sons order */
   const int *sons;
   const int *label;     /* nodes[i].label (branch type), [n_nodes] */
   const unsigned char *scale_node; /* com.nodeScale[n_nodes] or NULL (treesub.c:7177) */
   const unsigned char *z;          /* com.z[tip][h], row-major [n_tips][n_patt] */
   int cleandata;        /* com.cleandata */
   int n_codes;          /* rows of the ambiguity map in use */
   const int *n_chara;   /* nChara[code]      (tools.c:20) */
   const unsigned char *chara_map; /* CharaMap[code][k], row-major [n_codes][n] */
   const double *weights;/* com.fpatt[n_patt] */
   int n_genes;          /* com.ngene */
   const int *gene_off;  /* com.posG[n_genes+1] */
   const double *gene_rate; /* com.rgene[n_genes] */
   int n_pi;             /* 1 or n_genes */
   const double *pi;     /* com.pi (or per gene under Mgene): [n_pi][n] */
   int n_eigen;
   const orc_eigen *eigen;
   int mode;             /* ORC_MODE_LFUN (treesub.c:7764) or ORC_MODE_LFUNDG (treesub.c:7608) */
   int K;                /* com.ncatG (1 under lfun) */
   const double *freqK;  /* com.freqK[K] */
   const double *rate;   /* _rateSite per class (treesub.c:7669 / 7678) [K] */
   int n_labels;
   const int *eigen_of;  /* eigen set for (gene, class, label): [n_genes][K][n_labels] */
   const double *qfactor;/* Qfactor for (class, label): [K][n_labels] (treesub.c:7549, 7587) */
   const double *branch; /* nodes[i].branch, [n_nodes] (root entry unused) */
   long z_stride;        /* row stride of z; 0 = n_patt */
   int rate_gs;          /* Malpha (a gamma shape per gene, SetPGene baseml.c:1460): rate is [n_genes][K] and this is K; else 0 */
} orc_problem;

/* P(t) builders */
void orc_pmat_uvroot(double *P, double t, int n, const double *U, const double *V, const double *Root);
void orc_pmat_cijk(double *P, double t, int n, int nR, const double *Cijk, const double *Root);
void orc_pmat_k80(double *P, double t, double kappa);
void orc_pmat_jc69like(double *P, double t, int n);
void orc_pmat_qmat(double *P, double t, int n, const double *Q);   /* UNREST: matexp(Qt, n, 7, 5), n <= 8 */
/* GetPMatBranch restated: P(t) for the branch above `node`, for (gene, class). */
void orc_pmat_branch(const orc_problem *pb, int gene, int iclass, int node, double *P);

/* One com.plfun call.  Returns +lnL.  Optional outputs (NULL to skip):
 *   lnf[n_patt]            per-pattern log f_h as print_lnf_site would print it (treesub.c:7598)
 *   fhK[K*n_patt]          com.fhK (class-major) as fx_r leaves it (treesub.c:7743-7749)
 *   partials               nodes[].conP for all internal nodes and classes:
 *                          [K][n_nodes-n_tips][n_patt][n]  (PointconPnodes treesub.c:3518, conPSiteClass layout 7717)
 *   scalef                 com.nodeScaleF: [K][n_scale][n_patt] (treesub.c:7207-7227)
 * nthreads>1 shards the pattern loops with OpenMP (the reference itself is single-threaded).
 */
double orc_eval(const orc_problem *pb, double *lnf, double *fhK, double *partials, double *scalef, int nthreads);

/* lfunAdG (treesub.c:7447): forward pass of the rate chain MK[K*K] over the ls sites in their original order
 * (pose[site] = pattern index) on top of fx_r's fhK.  Returns +lnL. */
double orc_eval_adg(const orc_problem *pb, const double *MK, const int *pose, int ls);

/* orc_eval over pattern blocks spread over nthreads host cores (each thread walks the whole tree for its block; single
 * gene).  Returns +lnL. */
double orc_eval_blocked(const orc_problem *pb, int nthreads, int block);

/* Branch-local log-likelihood and its first two derivatives in the branch length, for the branch above `node_b`
 * (father a), at each of the n_t trial lengths t[]: restates lfuntdd / lfuntdd_SiteClass (treesub.c:8204-8296,
 * 8403-8541; lfunt / lfunt_SiteClass 8127, 8298 are the l-only special case):
 *     f_h = sum_ir freqK_ir sum_i pi_i B_i[h] sum_j P_ij(t) A_j[h],   P, dP, ddP = sum_k U[:,k] e^{t mu_k} {1, mu_k, mu_k^2} V[k,:]
 * with mu_k = rgene * Root_k * rateSite_ir * Qfactor (plain exp, e^{t mu_0} forced to 1, no clamp), A = partial of a
 * looking away from b, B = partial of b's subtree (or the state set of tip b).  The reference gets A by re-rooting at b
 * (ReRootTree treespace.c:236 + updateconP treesub.c:7982); here A and B are computed directly as the two messages
 * across the branch, which is the same product of the same P(t) factors.  Outputs are in lnL convention
 * (lnL, dlnL/dt, d2lnL/dt2 = -l, -dl, -ddl of the reference).  scale_node flags are ignored (scaling changes no value,
 * only the exponent range).  Returns 0 on success. */
int orc_eval_branch(const orc_problem *pb, int node_b, int n_t, const double *t, double *lnL, double *dlnL, double *ddlnL);

/* Marginal posterior probabilities of the states at internal node `node`, post[n_patt][n] (PostProbNode treesub.c:6142 after
 * re-rooting at the node; reversible models). */
int orc_node_posterior(const orc_problem *pb, int node, double *post);

/* Number of (branch, class) P(t) constructions performed by the last orc_eval (mirrors NPMatUVRoot, tools.c:88). */
long orc_last_npmat(void);

#ifdef __cplusplus
}
#endif
#endif
