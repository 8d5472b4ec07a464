/* internal definitions of the host library */
#ifndef PAMLH_INTERNAL_H
#define PAMLH_INTERNAL_H
#include "../../include/paml_amd.h"
#include "../../include/pamlh.h"

#define PAMLH_MAXOPT 64
#define PAMLH_MAXGENE 16
enum { JC69, K80, F81, F84, HKY85, T92, TN93, REV, UNREST };   /* baseml models (baseml.ctl) */

typedef struct {
   char key[PAMLH_MAXOPT][32], val[PAMLH_MAXOPT][1024];   /* up to PAMLH_MAXOPT `key = value` lines */
   int n;
} pamlh_ctl;

#define PAMLH_MAXEIG 512      /* eigen systems: site classes x branch types, or one per node (nhomo >= 2) */
typedef struct {
   int kind, nR;
   double kappa;
   double *U, *V, *Root, *Cijk;
   /* a reversible rate matrix whose decomposition is left to the device (paml_amd_set_eigen_qrev_batch): Q[n*n], its frequencies
    * and the scale Root is divided by; `lazy`: U, V, Root above have not been formed on the host (pamlh_eig_host does it on demand) */
   double *Q, *qpi, scale;
   int lazy;
   /* a codon matrix's elements at the positions it can have (codon_pairs: what goes to paml_amd_set_eigen_qrev_batch_sparse), picked out
    * by the thread that built Q, while it is in its cache */
   double *qv;
   int qv_n;
} pamlh_eig;

/* decompositions waiting for ONE device call: the eigen systems of all the candidates of a batched evaluation */
typedef struct {
   int n, cnt, cap;
   int *ids;
   double *Q, *pi, *scale;
   /* codon matrices travel as the elements they can have (paml_amd_set_eigen_qrev_batch_sparse): Q then holds [cnt][nnz] values */
   int nnz;
   const int *row, *col;
} pamlh_eig_batch;

struct pamlh {
   char err[512], dir[1024];
   int is_codeml;
   pamlh_ctl ctl;
   /* options */
   int seqtype, codonfreq, model, nssites, icode, fix_kappa, fix_omega, fix_alpha, ncatG, cleandata_opt, fix_blength, aa_model;
   double kappa0, omega0, alpha0;
   char seqfile[1024], treefile[1024], aaratefile[1024];
   /* data */
   int n, ns, ls, npatt, n_codes, cleandata;
   char **names;
   unsigned char *z;
   double *w;
   char *raw;              /* [ns][npatt*n31] raw characters of the patterns (for lnf output) */
   int n31;
   int *n_chara;
   int *pose, n_pose;      /* site (after cleaning) -> pattern index */
   int ngene, posG[PAMLH_MAXGENE + 1], lgene[PAMLH_MAXGENE];   /* option G: first pattern / number of sites of every gene */
   int nhomo;              /* baseml: 1 = base frequencies are parameters; 2 = a kappa per branch; 3, 4 = frequency sets (and kappas) per branch */
   int mg;                 /* CodonFreq 4 / 5: F1x4MG / F3x4MG */
   int fix_rho, adg;       /* auto-discrete-gamma: rho free or fixed != 0; adg: the current model state uses lfunAdG with MK */
   double rho0, rho, MK[64 * 64 / 4];
   int clock;              /* 1: global clock, x holds the internal node ages */
   int m2a_rel;            /* NSsites = 22 */
   int mgene;              /* Mgene: 0 rates, 2 different pi, 3 different kappa (& omega), 4 both */
   double piG[PAMLH_MAXGENE][64];   /* frequencies of every gene (com.piG) */
   double fb3x4G[PAMLH_MAXGENE][12], fb4G[PAMLH_MAXGENE][4];      /* position x nucleotide / nucleotide frequencies of every gene (the MG-style models, com.f3x4[igene]) */
   double rgene[PAMLH_MAXGENE];     /* com.rgene: rate of every gene relative to the first */
   int *gene_eigen_of;     /* [ngene][K][n_labels] when ngene > 1 */
   unsigned char *chara_map;
   double fb3x4[12], fb4[4], fcodon[64], pi_data[64];
   double fb61[64];
   int translate;          /* seqtype = 3: the sequence file holds codons, the analysis is of their amino acids */
   int free_ratio;         /* codeml model = 1: run as the branch model with every branch its own label */
   int itree, ntrees;      /* which tree of the tree file this analysis uses; how many the file holds */
   /* codon frequencies as parameters (estFreq = 1) and the mutation-selection models FMutSel0 / FMutSel (CodonFreq 6, 7) */
   int codonf_model, est_freq, npi, mutsel;      /* CodonFreq as given (0..7); estFreq; frequency parameters in x; 0 / 1 = FMutSel0 / 2 = FMutSel */
   double pf3x4[12], pi_aa[20];                  /* position x nucleotide frequencies in effect; observed amino-acid frequencies */      /* codon frequencies implied by the amino-acid frequencies (aa models 5, 6) */
   double aaS[400], aapi_file[20];
   /* tree */
   int nnode, root, nbranch;
   int *sons_ptr, *sons, *label, *branch_node, *father;
   double *tree_branch;    /* lengths read from the tree file, per node (-1: absent) */
   unsigned char *scale;
   /* model state */
   int np, ntime, mode, K, n_eigen, n_labels, n_pi;
   double *branch, *pi, *freqK, *rate;
   int *eigen_of;
   pamlh_eig eig[PAMLH_MAXEIG];
   double kappa, omega, alpha;
   double class_w[64];     /* NSsites: omega of every site class; branch model: omega of every label */
   char code[65];          /* genetic code: amino acid of each of the 64 codons (T, C, A, G order), '*' = stop */
   int n_omega;            /* branch model: number of branch labels = omegas */
   double qfactor[64];     /* branch-site / clade models: time scale of (class, branch type), [K][n_labels] */
   int use_qf;
   double ns_mr;           /* NSsites: mean rate at the mean omega = 1 / Qfactor_NS of the last pamlh_set_x */
   /* aaDist = 7 (AAClasses, codeml.c:4079 GetOmegaAA): dN/dS classes of amino-acid pairs, from OmegaAA.dat beside the ctl */
   int tipdate;                /* TipDate: the sequence names end in sampling dates; x has the mutation rate after the node ages */
   double tip_timeunit, *tip_age, *age_low;   /* ages of the tips (0 = the youngest) in time units; lowest possible age of every node */
   unsigned char aa1step[400];   /* aa models 8, 9 (REVaa_0 / REVaa): the exchangeabilities that are parameters (i > j), in x's order */
   int n_aarate;
   int *nh_label, nh_nbtype;   /* nhomo = 5 / fix_kappa = 2: the tree file's '#' labels name the frequency / rate sets; number of branch types */
   int *rate_label, n_brate;   /* clock = 2: rate class of the branch above every node, number of classes */
   int malpha;               /* Malpha: a gamma shape per gene; rate[] then holds [gene][class] */
   int opt_transformed;      /* pamlh_optimize is iterating on transformed proportions (pamlh_opt.c) */
   int aadist, n_omega_type;
   signed char omega_class[26][26];
   double aa_dist[26][26];           /* aaDist 1..6 / -1..-6: amino-acid distances over their maximum, by letters (GetDaa codeml.c:3967-3993) */
   double aa_omega[26][26];          /* omega of every amino-acid pair at the current parameters (GetOmega codeml.c:3020) */   /* by amino-acid letters: class of the pair, -1 = no one-step change under the code */
   /* optimiser state (pamlh_opt.c) */
   unsigned char *frozen;  /* NULL, or [np]: parameters pamlh_optimize leaves where they are (minB holds the branch lengths) */
   int opt_lean;           /* 1: fewer trial points per line search, no curvature pre-pass (the inner ming2 of minB) */
   double opt_abs_tol;     /* > 0 (lean mode): stop as soon as an iteration gains less than this in lnL (ming2's e) */
   /* pattern shard of a multi-GPU run (pamlh_set_shard): this process holds patterns [shard_first, shard_first + npatt) of npatt_global */
   int shard_rank, shard_world, shard_have_id;
   long npatt_global, shard_first;
   unsigned char shard_id[PAML_AMD_COMM_ID_BYTES];
   /* engine */
   paml_amd_engine *eng;
};

/* numerics (pamlh_num.c) */
int pamlh_simplex_groups(const pamlh *p, int *start, int *len, int cap);
int pamlh_nh_nrate(const pamlh *p);      /* nhomo >= 2: rate parameters, frequency sets (baseml.c:1201-1232) */
int pamlh_nh_npi(const pamlh *p);
void pamlh_eigen_sym(double *A, int n, double *w, double *R);
void pamlh_eigen_qrev(const double *Q, const double *pi, int n, double *Root, double *U, double *V);
void pamlh_eig_host(pamlh_eig *e, int n);
void pamlh_eig_release(pamlh_eig *e);
int pamlh_upload_eigen_sets(pamlh *p, paml_amd_engine *eng, int base, pamlh_eig_batch *batch);
int pamlh_eig_batch_flush(pamlh *p, paml_amd_engine *eng, pamlh_eig_batch *batch);
double pamlh_gammp(double a, double x);
double pamlh_quantile_gamma(double p, double alpha, double beta);
void pamlh_discrete_gamma(double *freqK, double *rK, double alpha, int K);
double pamlh_betai(double a, double b, double x);
double pamlh_quantile_normal(double prob);
double pamlh_cdf_normal(double x);
double pamlh_lbinormal(double h, double k, double r);
void pamlh_autod_gamma(double *M, double *freqK, double *rK, double alpha, double rho, int K);
double pamlh_quantile_beta(double prob, double p, double q);

/* io (pamlh_io.c) */
int pamlh_read_ctl(pamlh *p, const char *path);
const char *pamlh_opt(const pamlh *p, const char *key);
int pamlh_ctl_override(pamlh *p, const char *text);
const char *pamlh_genetic_code(int icode);      /* 64 characters, codons in T C A G order, '*' = stop; NULL: no such code */
double pamlh_optd(const pamlh *p, const char *key, double dflt);
int pamlh_read_seqs(pamlh *p);
int pamlh_read_tree(pamlh *p);
int pamlh_fail(pamlh *p, const char *fmt, ...);
int pamlh_force_host_eigen(void);
int pamlh_engine_ready(pamlh *p);
int pamlh_engine_model(pamlh *p);      /* pi, eigen systems and class tables of the current model state -> engine */
int pamlh_model_feasible(const pamlh *p);
int pamlh_x_to_branches(const pamlh *p, const double *x, double *branch);
pamlh *pamlh_state_clone(const pamlh *p);
void pamlh_state_free(pamlh *q);
#endif
