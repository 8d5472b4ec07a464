// kernel_args.h — argument structures of the hand-written kernels (kernels_*.h), visible to every translation unit of the
// engine: the kernels themselves are defined in exactly one (engine_eval.hip: P(t), pruning, reduction; engine_branch.hip:
// branch-local derivatives, node posteriors; engine_beb.hip: the BEB grid), the others launch them through the wrappers
// engine_state.h declares.
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

// Kernel A (kernels_pmat.h): batched P(t)
struct PmatArgs {
   int n, n_nodes, root, K, n_genes, n_labels, n_codes, layout;   // layout 0: VALU (row-major), 1: mfma64, 2: as 0 with the tip rows in m20 order, 3: as 1 with rows 48..60 of a 61-state P in the per-tree kernel's row-tail form
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
   int plain_codes;               // codes 0 .. plain_codes - 1 are single states equal to the code (pmat_mfma_kernel: no map needed for them)
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
   int npb;                       // nodes per workgroup of pmat_kernel_t (0 = 1)
   // single evaluations on pmat_mfma_kernel: what the chain label -> eigen_of -> eigen set -> U / V / Root resolves to, per (parameter
   // set, node), formed on the host when the tree, the class table or an eigen set changes — one table load in front of the matrix
   // loads instead of three dependent ones (a small-data evaluation is a chain of such round trips).  null: resolve in the kernel.
   const struct PmatRes *res;
};
struct PmatRes {
   const double *U, *V, *Root;
   double rate, qfactor;          // t = ((branch * rate) * gene rate) * qfactor: the kernel's own order
   int leaf, pad;
};

// Branch lengths and gene rates handed over INSIDE the kernel arguments (single evaluations of trees with up to ~440 nodes):
// the launch itself carries them, so an evaluation needs no host-to-device copy and no staging buffer to keep alive.
#define PMAT_INLINE_MAX 440
struct InlineVec {
   int n_branch, n_rate;          // 0, 0: read PmatArgs::branch / gene_rate instead
   double v[PMAT_INLINE_MAX];     // branch[n_branch], then gene_rate[n_rate]
};

// Kernel C (kernels_reduce.h)
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

// Branch-local evaluation (kernels_branch.h)
struct DerivArgs {
   int n, K, n_genes, n_labels, n_t, label, rate_gs;      // rate_gs: as PmatArgs
   const double *t;            // [n_t]
   const double *rate, *gene_rate, *qfactor;
   const int *eigen_of;
   const EigenDev *eigen;
   double *out;                // [pset][n_t][3][n*n]
   double *frag;               // non-null: also [pset][n_t][3][4096], each matrix in MFMA A-operand order (as pmat_kernel's pint)
};

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

// Branch-local evaluation in the eigen basis (kernels_branch.h: branch_eigprep_kernel, branch_eig_kernel, branch_poly_kernel).
// With P(t) = U diag(e^{mu_k t}) V the contraction of lfuntdd factorises:
//   f(t) = sum_i pi_i B_i (P(t) A)_i = sum_k e^{mu_k t} z_k w_k,   w = V A,   z = U^T (pi o B),
// and f', f'' take mu_k e^{mu_k t}, mu_k^2 e^{mu_k t}: TWO matrix products per pattern whatever the number of trial lengths and
// derivatives (the P / dP / ddP form needs three per trial length), and the 64 coefficients c_k = z_k w_k per pattern and class,
// kept in HBM, serve every further trial length on the same branch with no matrix product at all.
#define BEIG_NT 4      /* trial lengths evaluated per launch (more: further launches of the polynomial kernel on the stored coefficients) */
struct EigPrepArgs {
   int n, K, n_labels, n_t, label, n_codes, rate_gs, only_etab;
   const double *t;                     // [n_t]
   const double *rate, *gene_rate, *qfactor, *pi;      // pi: plain [n]
   const int *eigen_of;                 // [K][n_labels] (one gene)
   const EigenDev *eigen;
   const unsigned long long *code_mask; // [n_codes]
   double *efrag;                       // [K][2][4096]: V, then U^T diag(pi), both in MFMA A-operand order (rows = eigen index k)
   double *ztab;                        // [K][n_codes][64]: z of a tip's code, element q*16 + m = z_{4m+q}
   double *etab;                        // [K][n_t][3][64]: e^{mu t}, mu e^{mu t}, mu^2 e^{mu t}, element q*16 + m = k = 4m+q (k = 0: 1, 0, 0)
   double *ecol;                        // 61 states: [K][2][64] column 60 of V and of U^T diag(pi), element q*16 + m = row 4m+q (null: not wanted)
};

struct BranchEigArgs {
   int n, K, n_patt, n_tips, n_int, n_nodes, n_groups, n_scale, n_t, n_codes;
   int a_node, b_node;                  // the branch's two ends; b may be a tip
   int n_sons, son[2];                  // n_sons > 0: A's partial is formed here from its sons in the tree seen from the branch (then stored)
   int no_store;                        // timing experiment: the coefficients are not written
   int feval;                           // K == 1: lnL, dlnL, ddlnL of the n_t (<= BEIG_NT) trial lengths are formed in the same pass
   int chunk_groups, nb_local, first_chunk, n_out;      // partial sums [n_out = 3 n_t columns][n_rows]: a row per wave's eighth of a chunk of 16 * chunk_groups patterns, at global positions
   long n_rows;
   double *partials;                    // [K][n_int][n_groups][1024]  (read; A's slot written when n_sons > 0)
   const double *scalef;                // [K][n_scale][n_patt] or null
   const unsigned char *z;              // [n_tips][n_patt]
   const double *pint;                  // [K][n_nodes][4096]
   const double *ptip;                  // [K][n_nodes][tip_words]
   long tip_words;
   const double *efrag, *ztab, *etab;
   const double *ecol, *pcol;           // 61 states: column 60 of V / U^T diag(pi) ([K][2][64]) and of every P ([K][n_nodes][64])
   const double *freqK, *weights;
   double *coef;                        // [K][n_groups][1024]: c_k (x freqK x the class's scale factor relative to the pattern's largest)
   double *partial;                     // [nb_global][n_out]
};

struct BranchPolyArgs {
   int K, n_patt, n_groups, n_scale, n_t, it0, nt_here;      // this launch: trial lengths it0 .. it0 + nt_here - 1 of n_t
   int chunk_groups, nb_local, first_chunk, n_out;
   long n_rows;
   const double *coef, *etab, *scalef, *weights;
   double *partial;
};

// Node posteriors (kernels_branch.h)
struct PostArgs {
   int n, K, n_genes, n_patt, n_pi;
   const double *L, *S;        // [K][n_patt][n], summed scale factors [K][n_patt] or null
   const double *pi, *freqK;
   const int *gene_off;
   double *post;               // [n_patt][n]
};

// BEB grid integral (kernels_beb.h)
#define BEB_MAXK 32
#define BEB_MAXCLS 8
struct BebArgs {
   int n_patt, K, n_grid, n_cls, n_pblk, patt_per_blk;
   int log_form;             // fhK holds logarithms (trees with scaling nodes)
   int phase;                // beb_finish: 0 = sums and weights; 1 = the shard's sums only; 2 = weights from sums (all-reduced over the ranks in between)
   const double *fhK, *weights;
   double *f;                // [K][n_patt] scaled copy
   const double *pcl;        // [n_grid][n_cls]
   const int *iw;            // [n_grid][n_cls]
   const double *w_class;    // [K]
   double *part;             // [n_grid][n_pblk]
   double *lnfxs, *wg, *fx;  // [n_grid], [n_grid], [1]
   double *pr_last, *mean_w, *sd_w;   // [n_patt]
};

}  // namespace paml_amd
