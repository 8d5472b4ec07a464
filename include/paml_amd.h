/* paml_amd.h — C ABI of the MI355X-native likelihood engine (libpaml_amd.so).
 *
 * This is the drop-in boundary for PAML's likelihood hot path.  The reference has no plugin/FFI
 * layer; its de-facto seams are plain C symbols + globals:
 *     double (*com.plfun)(double x[], int np)           codeml.c:125  / baseml.c:70
 *     int ConditionalPNode(int inode, int igene, double x[])   codeml.c:3526 / baseml.c:1517
 *     int GetPMatBranch(double Pt[], double x[], double t, int inode)   treesub.c:7503 / 7534
 *     int PMatUVRoot(double P[], double t, int n, double U[], double V[], double Root[])  tools.c:516
 * Each entry point below names the reference interface (file:line under /root/reference/src) whose
 * role it takes over.  All pointers are host pointers unless the name starts with d_.  Every call
 * returns 0 on success or a negative PAML_AMD_E* code (the engine never exits the process, unlike
 * zerror() tools.c:1204); paml_amd_last_error() gives the message.
 *
 * Threading: one engine = one GPU + one HIP stream; calls on one engine must be serialised by the
 * caller, different engines (also on different devices and host threads) are independent.
 * Multi-GPU = one process and one engine per GPU over a contiguous pattern shard; after
 * paml_amd_comm_init every evaluation entry point returns the TOTAL over all ranks (the one
 * exchange step, an RCCL all-reduce over xGMI, runs inside the call on the engine's stream) — as the
 * reference's one com.plfun call returns the lnL of the whole alignment (codeml.c:748).
 */
#ifndef PAML_AMD_H
#define PAML_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct paml_amd_engine paml_amd_engine;

enum {
   PAML_AMD_OK = 0,
   PAML_AMD_EINVAL = -1,    /* bad argument / call order */
   PAML_AMD_ENOMEM = -2,    /* device or host allocation failed */
   PAML_AMD_EHIP = -3,      /* HIP runtime error (no device, launch failure ...) */
   PAML_AMD_EUNSUPPORTED = -4,
   PAML_AMD_ENOCONV = -5    /* a device eigen-decomposition (paml_amd_set_eigen_qrev_batch) hit its sweep limit: reported by the next
                               evaluation; decompose on the host and pass U, V, Root (paml_amd_set_eigen_uvroot) */
};

/* create flags */
enum {
   PAML_AMD_KEEP_PARTIALS = 1,  /* keep every internal node's partials resident: needed by
                                   paml_amd_get_partials / dirty-node re-evaluation
                                   (com.conPSiteClass = 1 memory model, codeml.c:2343-2346) */
   PAML_AMD_JIT = 2,            /* compile a pruning kernel specialised for the tree (hiprtc, a few seconds per
                                   new topology) even for small data sets; large ones do so by default
                                   (env PAML_AMD_JIT=0/1 overrides) */
   PAML_AMD_SHARD = 4           /* this engine will hold one rank's SHARD of a larger alignment (paml_amd_comm_init with
                                   n_patt_global > n_patt): kernels whose summation order differs are then never chosen by
                                   the shard's own size, so that lnL keeps the same bits for every number of ranks.  (Only
                                   20-state data sets of at most 4096 patterns are affected: without the flag they take the
                                   latency path of small data; paml_amd_comm_init refuses such an engine as a shard.) */
};

/* eigen-system kinds = the branches of GetPMatBranch (treesub.c:7503-7592) */
enum { PAML_AMD_EIGEN_UVROOT = 0, PAML_AMD_EIGEN_CIJK = 1, PAML_AMD_EIGEN_K80 = 2, PAML_AMD_EIGEN_JC69LIKE = 3, PAML_AMD_EIGEN_QMAT = 4 };

/* likelihood reduction = which com.plfun the reference would have installed (codeml.c:2338-2340) */
enum { PAML_AMD_MODE_LFUN = 0 /* treesub.c:7764 */, PAML_AMD_MODE_LFUNDG = 1 /* treesub.c:7608 + fx_r 7696 */ };

/* Once per data set, after the sequence file is read (replaces the conP/fhK/nodeScaleF allocations
 * of GetInitials codeml.c:2330-2354 and PointconPnodes treesub.c:3518).  n_states = com.ncode (<= 64),
 * n_tips = com.ns, n_patt = com.npatt, max_classes >= com.ncatG, n_genes = com.ngene.
 * The engine is created on the calling thread's current HIP device. */
int paml_amd_create(paml_amd_engine **out, int n_states, int n_tips, int n_patt, int max_classes, int n_genes,
                    unsigned flags);
void paml_amd_destroy(paml_amd_engine *e);
const char *paml_amd_last_error(const paml_amd_engine *e);

/* ---- Pattern shards over several GPUs (site patterns are independent given the tree, P(t), pi and the class
 * table; the reference has no counterpart — its lfun loops over all of com.npatt on one core, treesub.c:7764-7800).
 *
 * paml_amd_shard_bounds (host only, no GPU): the contiguous block [*first, *first + *count) of n_patt_global patterns that
 *   rank `rank` of `world` owns.  Blocks are cut at multiples of the reduction chunk (a function of n_patt_global alone,
 *   >= 256 patterns), so the partial sums of every rank are entries of ONE global array.  More ranks than chunks
 *   (paml_amd_max_ranks) would leave ranks without patterns: PAML_AMD_EUNSUPPORTED, for every rank alike, so that the ranks of
 *   a job fail together before any of them enters a collective call.
 * paml_amd_comm_unique_id: rank 0 obtains the 128-byte RCCL id (ncclGetUniqueId) and hands it to the other ranks by any
 *   means (a pipe, a file, MPI, torch.distributed's store).
 * paml_amd_comm_init: collective over the ranks (ncclCommInitRank).  The engine was created for its shard's n_patt and holds
 *   patterns [first_pattern, first_pattern + n_patt) of n_patt_global.  From then on eval / eval_device / eval_batch /
 *   eval_dirty / eval_branch return totals over all ranks, identical bits on every rank; the lnL of eval* is moreover
 *   independent of `world` (the ranks' zero-padded partial-sum arrays are added — exact — and summed in one fixed order).
 *   lnf / fhK / partials / posteriors stay per-shard.  world = 1 with a non-NULL id makes a one-rank communicator (the
 *   collective path on a single GPU); world = 1 with id = NULL only sets the global chunking.
 *   eval_adg (the rate chain runs over the sites in order): pose[] holds GLOBAL pattern indices, the same on every rank; the
 *   shards' class likelihoods are gathered over the ranks and every rank runs the chain — the same bits as a single engine.
 *   beb_grid / beb_grid_classes: a grid point's log-likelihood is a sum over all patterns, so the shards' sums are all-reduced
 *   before the grid weights are formed; every rank passes the same grid and gets the posteriors of ITS patterns and the
 *   global ln_fx.  node_posterior and the host's NEB work per pattern and need no exchange.
 *   The exchange step (all-reduce of the partial sums + their fixed-order total) runs on a stream of the engine's own, ordered
 *   by events: consecutive paml_amd_eval_device calls prune evaluation i + 1 while evaluation i is being reduced.  Every other
 *   entry point, and paml_amd_flush, joins the engine's stream to the totals still on their way.
 * The RCCL library (librccl.so.1, or the one PAML_AMD_RCCL_LIB names) is bound at run time on first use; PAML_AMD_EUNSUPPORTED
 * when it is not installed. */
/* GPUs visible to this process, and the one the calling thread's next paml_amd_create uses (hipSetDevice): a host written in C
 * — one process per GPU, as `pamlh_lnl --gpus N` forks them — needs no HIP headers. */
int paml_amd_device_count(void);
int paml_amd_set_device(int device);
#define PAML_AMD_COMM_ID_BYTES 128
int paml_amd_shard_bounds(long n_patt_global, int world, int rank, long *first, long *count);
int paml_amd_max_ranks(long n_patt_global);      /* the number of reduction chunks = the largest world shard_bounds accepts */
int paml_amd_comm_unique_id(void *id128);
int paml_amd_comm_init(paml_amd_engine *e, int rank, int world, const void *id128, long n_patt_global, long first_pattern);
int paml_amd_comm_destroy(paml_amd_engine *e);
int paml_amd_comm_info(const paml_amd_engine *e, int *rank, int *world, long *n_patt_global, long *first_pattern, int *chunk);
/* Which collective library the engine's exchange step is bound to (no reference counterpart): the path of the shared object that
 * holds ncclAllReduce (dladdr), loading it first if no communicator has been made yet — PyTorch ships a librccl of its own, and a
 * process that has imported torch shares THAT copy (dlopen matches by soname).  Returns the length written (truncated to cap - 1),
 * PAML_AMD_EUNSUPPORTED when no library could be bound. */
int paml_amd_comm_library(char *path, int cap);
/* Diagnostics of the exchange step (no reference counterpart; what `bench.py --gpus N` prints so that a scaling run explains
 * itself).  enable != 0 switches timed events on for the evaluations that follow (also: PAML_AMD_COMM_STATS=1), 0 off.  With
 * non-NULL outputs it reports, over the last *n (<= 64) evaluations whose exchange step ran on the engine's collective stream
 * — the caller has flushed and synchronised the stream —
 *   exchange_us:  partial sums ready -> total formed (the all-reduce over the ranks + the fixed-order total), mean and maximum;
 *   lane_wait_us: how long an evaluation's pruning stream stood in front of the previous exchange of its slot (0 when the
 *                 exchange had finished before the stream got there), mean and maximum. */
int paml_amd_comm_stats(paml_amd_engine *e, int enable, int *n, double *exchange_us_mean, double *exchange_us_max,
                        double *lane_wait_us_mean, double *lane_wait_us_max);

/* Launch all work on this hipStream_t (default: the null stream). */
int paml_amd_set_stream(paml_amd_engine *e, void *hip_stream);

/* com.z (treesub.c:1116 EncodeSeqs), nChara/CharaMap (tools.c:20, treesub.c:1218), com.fpatt, com.posG.
 * z is row-major [n_tips][n_patt] one byte per character code; with cleandata != 0 codes are states
 * 0..n-1 and the map may be NULL; otherwise n_chara[code] states listed in chara_map[code*n_states + k].
 * gene_off has n_genes+1 non-decreasing entries from 0 to n_patt (NULL = one gene covering all patterns; a pattern shard of a
 * multi-GPU run may hold nothing of a gene: equal neighbours).  Uploaded once. */
int paml_amd_set_tips(paml_amd_engine *e, const unsigned char *z, int cleandata, int n_codes, const int *n_chara,
                      const unsigned char *chara_map, const double *weights, const int *gene_off);

/* nodes[].sons / nson / label (codeml.c:143-147) as CSR, tree.root, com.nodeScale (treesub.c:7177; NULL = none).
 * Tips are nodes 0..n_tips-1.  The root may be a tip ("young ancestor", codeml.c:3535) and nodes may
 * have any number of sons.  Replaces the recursion of ConditionalPNode by a flattened post-order
 * program; send again after ReRootTree (treespace.c:236). */
int paml_amd_set_tree(paml_amd_engine *e, int n_nodes, int root, const int *sons_ptr, const int *sons,
                      const int *label, const unsigned char *scale_node);

/* com.pi (n_pi = 1) or com.piG per gene (n_pi = n_genes), row-major [n_pi][n_states]. */
int paml_amd_set_pi(paml_amd_engine *e, int n_pi, const double *pi);

/* The same decomposition done ON THE DEVICE, for a batch of reversible rate matrices at once — what eigenQREV (tools.c:5023-5110,
 * called by eigenQcodon codeml.c:3229 for every omega class of every trial point) does on one CPU core, 0.3-0.9 ms per 61 x 61
 * matrix.  Q[n_sets][n*n] row-major (only the lower triangle is read, like eigenQREV), pi[n_sets][n], scale[n_sets] (NULL = 1):
 * set set_ids[i] becomes Root = w / scale (descending), U = diag(1/sqrt pi) R, V = R^T diag(sqrt pi) with
 * diag(sqrt pi) Q diag(1/sqrt pi) = R diag(w) R^T; states with pi = 0 are left out (Root 0, unit rows / columns).  One workgroup
 * per matrix (cyclic Jacobi in LDS, FP64, ~0.5 ms cold): a gradient's or a line search's several hundred decompositions take about the time
 * of one, and U, V, Root never cross PCIe.  Asynchronous on the engine's stream.  paml_amd_get_eigen reads a set back (parity);
 * paml_amd_eigen_counters: matrices decomposed so far and the Jacobi sweeps each matrix of the last batch took (-1: the limit of 40
 * sweeps was reached without convergence — decompose that matrix on the host instead).  The decomposition is asynchronous, so a set
 * that did not converge is reported by the next synchronous evaluation (paml_amd_eval, _eval_batch, _eval_dirty, _eval_branch), which
 * returns PAML_AMD_ENOCONV instead of a likelihood formed from unconverged eigenvectors; the C host (pamlh_*.c) and the reference-side
 * binding (integration/codeml_plfun.patch) then decompose on the host and evaluate again. */
int paml_amd_set_eigen_qrev_batch(paml_amd_engine *e, int n_sets, const int *set_ids, const double *Q, const double *pi, const double *scale);
/* The same with the matrices handed over as the elements they can have: nnz positions (row[k] >= col[k]: the lower triangle and the
 * diagonal — what eigenQREV reads, tools.c:5048-5060 — each position once), shared by the batch, and vals[n_sets][nnz]; every other
 * element is zero.  A codon matrix has 263 elements below its diagonal (single-nucleotide changes between sense codons, universal
 * code) + 61 on it of 3 721: 2.6 KB instead of 30 KB per matrix cross from pageable host memory, which is what a batch call's host time
 * is (tools/eigen_call_cost.py).  Same decomposition; the stopping threshold's norm is summed in another order, so the last bits may
 * differ from the dense call's. */
int paml_amd_set_eigen_qrev_batch_sparse(paml_amd_engine *e, int n_sets, const int *set_ids, int nnz, const int *row, const int *col,
                                         const double *vals, const double *pi, const double *scale);
/* Warm start for the above (on = 1; 0 = off, the default; -1 = leave as is; *n_warm, if not NULL, receives the number of
 * decompositions that started warm so far): a decomposition starts its Jacobi sweeps from the eigenvectors of the NEAREST matrix any set
 * of the engine was last decomposed for (told by a signature of eight weighted row sums; round 6 — until then: the set's own previous
 * matrix) — the set's predecessor along a line search, the base point's sets for the perturbed points of a gradient: 2 sweeps after a
 * finite-difference step, 4 after a 5 % step, instead of 8-9 (0.52 ms per call cold, 0.19 ms after a finite-difference step, 0.31 ms
 * after a 5 % step: tools/eigen_probe.py).  The answer is a decomposition of the NEW matrix to the same threshold either way; its last
 * bits then depend on the matrices the engine held before, which is why the default is off.  Every 16th decomposition of a chain, and
 * any whose pi has other zero entries than every candidate's, starts cold. */
int paml_amd_set_eigen_warm_start(paml_amd_engine *e, int on, long *n_warm);
int paml_amd_get_eigen(paml_amd_engine *e, int set_id, double *U, double *V, double *Root);
int paml_amd_eigen_counters(paml_amd_engine *e, long *n_decomposed, int *sweeps_last_batch, int cap);

/* Eigen systems, mirroring U,V,Root / _UU,_VV,_Root[NBTYPE+2] (codeml.c:185, treesub.c:9250) and
 * Cijk,Root,nR (baseml.c:123-124).  set_id is dense from 0.  Small H2D copies, typically per evaluation. */
int paml_amd_set_eigen_uvroot(paml_amd_engine *e, int set_id, const double *U, const double *V, const double *Root);
int paml_amd_set_eigen_cijk(paml_amd_engine *e, int set_id, int nR, const double *Cijk, const double *Root);
int paml_amd_set_eigen_k80(paml_amd_engine *e, int set_id, double kappa);        /* PMatK80 tools.c:578 */
int paml_amd_set_eigen_jc69like(paml_amd_engine *e, int set_id);                 /* PMatJC69like codeml.c:3585 */
/* No eigen system at all: the rate matrix Q[n*n] itself (baseml UNREST / UNRESTu, QUNREST treesub.c:2543), from which
 * GetPMatBranch builds P(t) = e^{Qt} with matexp(Qt, n, 7 Taylor terms, 5 squarings) (treesub.c:7524-7526, tools.c:4879).
 * n_states <= 8.  The branch-local evaluation (eval_branch) is not available for this kind. */
int paml_amd_set_eigen_qmat(paml_amd_engine *e, int set_id, const double *Q);

/* Site classes: com.ncatG, com.freqK, per-class _rateSite (treesub.c:7669/7678), and for each
 * (gene, class, branch label) the eigen set (Set_UVR_BranchSite codeml.c:2663, SetPSiteClass
 * treesub.c:7663) and for each (class, label) the Qfactor applied to t (treesub.c:7549, 7587).
 * eigen_of is [n_genes][K][n_labels], qfactor is [K][n_labels] (NULL = all 1).  mode selects the
 * lfun (K must be 1) or the fx_r + lfundG reduction, including their different f<=0 floors. */
int paml_amd_set_classes(paml_amd_engine *e, int mode, int K, const double *freqK, const double *rate, int n_labels,
                         const int *eigen_of, const double *qfactor);

/* Malpha = 1 (a gamma shape per gene: lfundG calls SetPGene for every gene, which runs DiscreteGamma with that gene's alpha,
 * treesub.c:7627-7629, codeml.c:2441 / baseml.c:1460): class rates per gene, rate[n_genes][K], replacing the rate[K] of the
 * last set_classes (which also switches back).  With it, the `rate` argument of paml_amd_eval_batch is [n_batch][n_genes][K]. */
int paml_amd_set_gene_class_rates(paml_amd_engine *e, const double *rate);

/* One com.plfun call (codeml.c:748): batched P(t) for every branch x class x gene, fused pruning,
 * root / class-mixture / weighted-log reduction.  branch[n_nodes] = nodes[].branch, gene_rate =
 * com.rgene (NULL = 1).  Returns +lnL (the reference returns -lnL).  lnf[n_patt] (log f_h as
 * print_lnf_site treesub.c:7598 prints it) and fhK[K*n_patt] (com.fhK, class-major) are optional. */
int paml_amd_eval(paml_amd_engine *e, const double *branch, const double *gene_rate, double *lnL, double *lnf,
                  double *fhK);

/* Same evaluation without the final device->host copy or synchronisation: lnL (the total over the ranks when the engine has a
 * communicator) is left in d_lnL (a device pointer, e.g. a torch tensor).  Consecutive calls are pipelined: the next
 * evaluations' P(t) are built on a side stream, the pruning kernels of consecutive evaluations (large problems on the
 * matrix-core kernels) alternate between the engine's stream and a second stream of the engine's own so that one takes the CUs
 * as the other releases them, and the fixed-order total — with a communicator the exchange step in front of it — runs on a
 * third.  The values of a run are written in call order, but the last of them may still be on the engine's own streams:
 * paml_amd_flush makes the engine's stream wait for them.  Call it once after a run of eval_device calls, before synchronising
 * the stream or reading d_lnL on it (a device-wide synchronisation covers the engine's streams too); every other entry point
 * of the engine does it implicitly.  Give every evaluation of a run its own d_lnL slot if all values are wanted. */
int paml_amd_eval_device(paml_amd_engine *e, const double *branch, const double *gene_rate, double *d_lnL);
int paml_amd_flush(paml_amd_engine *e);
/* The device eigen-decompositions queued so far (paml_amd_set_eigen_qrev_batch): waits for the engine's stream and returns
 * PAML_AMD_ENOCONV if one of them reached its sweep limit — what every synchronous entry point (eval, eval_batch, eval_dirty, eval_branch,
 * eval_adg, beb_grid*, node_posterior) reports by itself; for callers of eval_device, which never synchronises.  paml_amd_flush reports
 * a failure that has already been recorded, without waiting. */
int paml_amd_eigen_status(paml_amd_engine *e);

/* Re-evaluate after a change that leaves the partials of the nodes with clean[node] != 0 valid
 * (com.oldconP, codeml.c:112, treespace.c:250): those subtrees are read back instead of recomputed.
 * Needs PAML_AMD_KEEP_PARTIALS and one earlier full evaluation with the same classes. */
int paml_amd_eval_dirty(paml_amd_engine *e, const double *branch, const double *gene_rate, const unsigned char *clean,
                        double *lnL);

/* lfunAdG (treesub.c:7447-7494), the auto-discrete-gamma model (rho != 0): fx_r runs on the device exactly as for lfundG;
 * the K-state rate chain with transition matrix MK[K*K] (AutodGamma tools.c:2630, the caller's) is then run over the ls
 * sites in their original order, pose[site] = pattern index (com.pose) — a sequential recurrence, done on the host from the
 * device's fhK.  Needs PAML_AMD_MODE_LFUNDG classes.  Returns +lnL. */
int paml_amd_eval_adg(paml_amd_engine *e, const double *branch, const double *gene_rate, const double *MK, const int *pose, int ls,
                      double *lnL);

/* The grid integral of Bayes empirical Bayes (lfunNSsites_M2M8 codeml.c:6482-6580; also the M8 / M2a tables of
 * get_pclassM_iw_M2M8, 6340) over the class likelihoods fhK of the LAST evaluation (lfundG mode, K <= 32 classes = all the
 * omega values the grid needs).  Grid point g is a mixture of n_cls classes: proportion pcl[g][c], class iw[g][c] (index into
 * the K classes); w_class[K] = omega of each class.  For every pattern: pr_last = posterior probability of the last class of
 * the mixtures, mean_w, sd_w = posterior mean and sd of omega; ln_fx (may be NULL) = log of the marginal likelihood over the
 * grid (up to the per-pattern scaling constants).  n_grid x n_patt x n_cls terms with a log each — the reference's second
 * hot loop at scale. */
int paml_amd_beb_grid(paml_amd_engine *e, int n_grid, int n_cls, const double *pcl, const int *iw, const double *w_class,
                      double *ln_fx, double *pr_last, double *mean_w, double *sd_w);

/* The same grid integral returning the posterior of EVERY mixture class — what lfunNSsites_ACD (codeml.c:6827-7010) computes
 * for branch-site model A (4 classes per grid point drawn from 121 evaluated (background, foreground) omega pairs, table 1 of
 * Yang, Wong & Nielsen 2005) and the clade models.  Any K (the class likelihoods are read from memory), n_cls <= 8.
 * post[n_cls][n_patt]; ln_fx as above. */
int paml_amd_beb_grid_classes(paml_amd_engine *e, int n_grid, int n_cls, const double *pcl, const int *iw, double *ln_fx, double *post);

/* n_batch evaluations in one launch: the finite-difference loops of the optimiser (gradientB tools.c:6561, the forward /
 * central differences of ming2 tools.c:6595 and of the Hessian, HessianSKT2004 treesub.c:7241) call com.plfun np (or 2np,
 * np^2) times on the same data with one parameter nudged; here those calls become the elements of one batch.  Element b
 * uses branch[b][n_nodes] and gene_rate[b][n_genes] (NULL = all 1) and, where a non-NULL table is given, its own
 * eigen_of[b][n_genes][K][n_labels], qfactor[b][K][n_labels], freqK[b][K], rate[b][K] — layouts of one element as in
 * paml_amd_set_classes; NULL = the tables set there, shared by all elements.  Eigen systems are referenced by set id, so
 * a nudged kappa / omega is a further paml_amd_set_eigen_* id.  lnL[n_batch] comes back (+lnL each), and, when lnf is
 * not NULL, the per-pattern log f_h of every element, lnf[n_batch][n_patt] — what HessianSKT2004 (treesub.c:7241) collects
 * in dfsites for its 2 np perturbed evaluations.  pi and the mode (lfun / lfundG) are those of the engine.  Not with
 * PAML_AMD_KEEP_PARTIALS. */
int paml_amd_eval_batch(paml_amd_engine *e, int n_batch, const double *branch, const double *gene_rate, const int *eigen_of,
                        const double *qfactor, const double *freqK, const double *rate, double *lnL, double *lnf);

/* Branch-local evaluation = lfuntdd / lfuntdd_SiteClass (treesub.c:8204, 8403; lfunt / lfunt_SiteClass 8127, 8298 are
 * the lnL-only case), the function minbranches (treesub.c:8039) iterates with Newton steps: for the branch above
 * node_b, and for each of the n_t (<= 64) trial lengths t[], the log-likelihood and its first two derivatives in t,
 * all other branch lengths taken from branch[].  Outputs are in lnL convention: lnL = -l, dlnL = -dl, ddlnL = -ddl
 * of the reference.  Like the reference (ReRootTree treespace.c:236 marks the nodes on the path, com.oldconP[a] = 0 at
 * line 250; updateconP treesub.c:7982 recomputes only those), the engine keeps the partials of BOTH sides of every edge
 * it has visited resident in HBM between calls: moving to a neighbouring branch recomputes the one or two nodes on the
 * path, a changed branch length only the partials that look across it.  The contraction reads the two resident partials;
 * for 21..64 states with (U, V, Root) eigen systems and one gene it works in the eigen basis (f(t) = sum_k e^{mu_k t} z_k w_k,
 * w = V A, z = U^T (pi o B): two matrix-core products per pattern for any number of trial lengths and derivatives, the
 * coefficients z_k w_k kept for further calls on the same branch), otherwise it builds P, dP, ddP for all trial lengths in one
 * batched kernel as lfuntdd does.  One host synchronisation per call.  Scaling nodes keep rescaling; their factors are stored
 * with the partials.
 * Any set_tips / set_tree / set_eigen_* / set_classes call drops the resident partials. */
int paml_amd_eval_branch(paml_amd_engine *e, int node_b, int n_t, const double *t, const double *branch,
                         const double *gene_rate, double *lnL, double *dlnL, double *ddlnL);
/* Work done by paml_amd_eval_branch so far: calls, and internal-node partials recomputed (a full tree costs n_nodes - n_tips). */
int paml_amd_branch_counters(const paml_amd_engine *e, long *n_calls, long *n_nodes_recomputed);
/* eval_branch calls so far that were served from the stored eigen-basis coefficients of the branch (21..64 states: a further
 * trial length on the branch just evaluated needs no matrix product; kernels_branch.h). */
long paml_amd_branch_coef_hits(const paml_amd_engine *e);
/* eval_branch calls so far in which most of the tree was dirty (every branch length moved: minB's round after ming2, treesub.c:7982) and
 * the forest of dirty subtrees ran on a per-tree kernel of its own instead of the interpreter (compiled in the background from the second
 * request of the same forest on; at once with PAML_AMD_JIT). */
long paml_amd_branch_refill_kernels(const paml_amd_engine *e);
/* With paml_amd_profile(e, 1): milliseconds (HIP events on the engine's stream) between the first and the last contraction kernel of
 * the last eval_branch call — the coefficient-forming kernel and / or the polynomial kernel(s); < 0 when the call took the
 * P / dP / ddP form or profiling was off. */
double paml_amd_branch_kernel_ms(paml_amd_engine *e);
/* Parity accessor: the per-block partial sums of the last eval_branch, [rows][cols = 3 n_t] at their global block positions
 * (after the all-reduce when a communicator is attached) — summed in a fixed order they give lnL, dlnL, ddlnL with the same bits
 * for every number of ranks.  out = NULL: the shape only. */
int paml_amd_get_branch_partials(paml_amd_engine *e, double *out, long cap, long *rows, int *cols);

/* Marginal ancestral reconstruction (AncestralMarginal treesub.c:6288, PostProbNode 6142): post[n_patt][n_states] =
 * posterior probabilities of the states at internal node `node` given the data, at the current classes / eigen systems and
 * the given branch lengths.  The reference re-roots the tree at the node and reuses conP (ReRootTree + updateconP); the
 * engine walks the tree rooted at the node in one fused pass.  Reversible models only. */
int paml_amd_node_posterior(paml_amd_engine *e, int node, const double *branch, const double *gene_rate, double *post);

/* Parity / post-processing accessors.
 * get_pmat: the matrix GetPMatBranch (treesub.c:7534) would have produced for the branch above
 *   `node`, row-major P[from*n + to], from the last evaluation.
 * get_partials: nodes[node].conP for class iclass, layout [n_patt][n_states] (PointconPnodes
 *   treesub.c:3518); needs PAML_AMD_KEEP_PARTIALS.
 * get_scale: com.nodeScaleF row of a scaling node, [n_patt] (treesub.c:7207-7227). */
int paml_amd_get_pmat(paml_amd_engine *e, int gene, int iclass, int node, double *P);
int paml_amd_get_partials(paml_amd_engine *e, int node, int iclass, double *conP);
int paml_amd_get_scale(paml_amd_engine *e, int node, int iclass, double *scale);
/* The partial sums sum_h w_h log f_h of the last evaluation, one per reduction chunk of the GLOBAL pattern range (zero for
 * chunks other ranks own when there is no communicator; the all-reduced array when there is one): the quantities of lfun's
 * accumulation loop (treesub.c:7796-7800) before the final total.  Returns the number of chunks. */
int paml_amd_get_partial_sums(paml_amd_engine *e, double *out, int cap);

/* Per-kernel timing with HIP events on the engine's stream (bench.py roofline leg).
 * profile(1) starts recording an event pair around every kernel launch; profile_read sums the
 * elapsed milliseconds per kernel family since the last read and resets the accumulators. */
int paml_amd_profile(paml_amd_engine *e, int enable);
int paml_amd_profile_read(paml_amd_engine *e, double *ms_pmat, double *ms_prune, double *ms_reduce, long *n_evals);

/* Counters mirroring NFunCall / NPMatUVRoot (tools.c:88, printed codeml.c:770). */
int paml_amd_counters(const paml_amd_engine *e, long *n_eval, long *n_pmat);

/* Host-only introspection (no GPU needed): the flattened post-order program the engine would run for
 * a tree — 4 ints per op (code, a, b, c; paml_amd/csrc/program.h).  Returns the number of ops (or a
 * negative error); at most cap ops are written.  Used by the CPU test-suite to check the traversal. */
int paml_amd_debug_program(int n_tips, int n_nodes, int root, const int *sons_ptr, const int *sons,
                           const unsigned char *scale_node, int keep_partials, const unsigned char *clean,
                           int *ops_out, int cap, int *max_stack);

/* Host-only: generate (and with compile != 0 also hiprtc-compile for gfx950, no GPU needed) the kernel
 * specialised for a tree (paml_amd/csrc/jit.h).  compile: bit 0 = compile; bits 8..15 = n_states (4, 5 or 20 select the
 * one-pattern-per-lane kernel, 64 + n the MFMA kernel trimmed to n states, anything else the 61-state MFMA kernel); bit 1 = the
 * fused 4 / 5-state kernel for K = bits 16..23 classes, bits 24..31 character codes and a reduction chunk of 256 x bits 2..7 patterns.  Returns the source length, or a negative error
 * with the compiler log in text_out. */
int paml_amd_debug_jit(int n_tips, int n_nodes, int root, const int *sons_ptr, const int *sons,
                       const unsigned char *scale_node, char *text_out, int cap, int compile);

/* Host-only (hiprtc cross-compiles without a GPU): compile the per-tree kernel an engine of these sizes would select for this tree
 * and keep the code object in `dir` (the library's read-only lib/jit directory, or a user cache), so that the first evaluation on
 * a fresh machine does not start with seconds of compilation.  n_patt_global fixes the reduction chunk the 4 / 5-state kernel is
 * specialised on; K = site classes. */
int paml_amd_jit_prebuild(int n_states, int n_tips, int n_codes, int K, long n_patt_global, int n_nodes, int root, const int *sons_ptr,
                          const int *sons, const unsigned char *scale_node, const char *dir, char *log_out, int log_cap);

/* Name of the pruning kernel the engine selected ("mfma64", "valu4", "valu20", ...). */
const char *paml_amd_kernel_name(const paml_amd_engine *e);

/* PatternWeight (treesub.c:1386-1516) on the device, stand-alone (no engine): collapse the n_sites alignment columns of
 * chars[n_seq][n_sites * width] (raw characters; width = 1, or 3 for codons) into distinct site patterns in the reference's
 * order — sorted by the characters of sequence 0, then sequence 1, ... (unsigned byte order = the strcmp order of the
 * reference's char + 1 strings), within genes when gene[n_sites] (com.pose on input) is given.  Out: *n_patt; first_site
 * [n_patt] = the first site showing each pattern (p2s[]), weights[n_patt] = com.fpatt, pose[n_sites] = pattern of every site
 * (com.pose on output).  first_site and weights need room for n_sites entries.  A radix sort of the site indices over the
 * key bytes, then run detection and a scan — HBM / latency-bound byte work.  PAML_AMD_EHIP when no device is visible. */
int paml_amd_compress_patterns(int n_seq, int n_sites, int width, const unsigned char *chars, const int *gene, int *n_patt,
                               int *first_site, double *weights, int *pose);

#ifdef __cplusplus
}
#endif
#endif
