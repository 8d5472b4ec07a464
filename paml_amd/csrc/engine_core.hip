// engine_core.hip — life cycle of the engine and its data: create / destroy, set_tips / set_tree / set_pi / set_eigen_* / set_classes,
// read-back of P(t), partials and scale factors (parity / debugging), stage timing, counters.
// Built for gfx950 only (one of the translation units of libpaml_amd.so, see engine_state.h).
#include "engine_state.h"
#include "eigen_kernels.h"

extern "C" {

int paml_amd_create(paml_amd_engine **out, int n_states, int n_tips, int n_patt, int max_classes, int n_genes,
                    unsigned flags)
{
   if (!out) return PAML_AMD_EINVAL;
   *out = nullptr;
   if (n_states < 2 || n_states > 64 || n_tips < 2 || n_patt < 1 || max_classes < 1 || n_genes < 1) return PAML_AMD_EINVAL;
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return PAML_AMD_EHIP;   // no CPU fallback, by design
   paml_amd_engine *e = new (std::nothrow) paml_amd_engine();
   if (!e) return PAML_AMD_ENOMEM;
   e->n = n_states; e->n_tips = n_tips; e->n_patt = n_patt; e->max_classes = max_classes; e->n_genes = n_genes;
   e->flags = flags;
   e->env.read();
   e->comm_stats = e->env.comm_stats;
   if (e->env.comm_cus >= 0) e->comm_cus = e->env.comm_cus;
   if (e->env.lanes) e->n_lanes = std::min(std::max(e->env.lanes, 2), (int)paml_amd_engine::MAXL);
   if (hipGetDevice(&e->device) != hipSuccess ||
       hipDeviceGetAttribute(&e->n_cu, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || e->n_cu < 1) {
      delete e;
      return PAML_AMD_EHIP;
   }
   e->n_patt_global = n_patt;
   e->chunk = red_chunk(n_patt);
   e->nb_global = (n_patt + e->chunk - 1) / e->chunk;
   {  // per-tree specialised kernels: on request, or by default once the data set is large enough to repay the compile
      const char *j = getenv("PAML_AMD_JIT");
      e->jit_enabled = (flags & PAML_AMD_JIT) != 0 || (j && j[0] == '1') || (!j && (long)n_patt * max_classes >= 65536);
      e->jit_forced = (flags & PAML_AMD_JIT) != 0 || (j && j[0] == '1');
      if (j && j[0] == '0') e->jit_enabled = false;
      const char *cj = getenv("PAML_AMD_COOPJIT");
      e->coopj_enabled = !(j && j[0] == '0') && !(cj && cj[0] == '0');
   }
   // 20 states: the specialised MFMA kernel trimmed to 2 row blocks x 5 k-blocks beats the scalar-operand kernel 2-3x; the
   // MFMA interpreters (64 MFMAs per product whatever n) do not, so small or keep-partials engines stay on valu20
   // 20 states: the per-tree kernel on v_mfma_f64_4x4x4 (no padding: 25 block products per 16 patterns) for one gene and trees whose
   // internal branches' P(t) fit in LDS; else the 16x16x4 kernel trimmed to 2 row blocks x 5 k-blocks (2-3x the scalar-operand
   // kernel); the MFMA interpreters (64 MFMAs per product whatever n) do not pay, so small or keep-partials engines stay on valu20
   // (round 5: trees with more internal branches than LDS holds P(t) blocks for — more than 49 taxa — read the others' operands from
   //  global memory: up to M20_MAX_TIPS taxa — measured at 60: 0.42 of the FP64 peak against the padded 16x16x4 kernel's 0.39; the
   //  packed tip codes of a unit live in registers, 64 VGPRs at 60 taxa, and beyond 64 taxa the walk would spill more than it gains)
   constexpr int M20_MAX_TIPS = 64;
   e->want_m20 = n_states == 20 && e->jit_enabled && !(flags & PAML_AMD_KEEP_PARTIALS) && n_tips <= (getenv("PAML_AMD_M20_49") ? 49 : M20_MAX_TIPS) && !e->env.no_m20 && !e->env.valu20;
   // ... and SMALL 20-state data sets (at most 4096 patterns, round 4): what counts there is the length of one wave's walk, and the
   // cooperative form of the MFMA interpreter (prune_mfma64_coop: four waves per 16-pattern group, 16 MFMAs per wave and branch on
   // the zero-padded matrices) walks a branch in a sixth of the time of the scalar-operand kernel's 400 dependent FMAs per lane
   // (stewart.aa, 6 taxa x 98 patterns x 4 classes: 40 -> 22 us per evaluation, profiles/r04_small_timeline.txt)
   // (its sums are ordered differently from the scalar-operand kernel's: an engine that holds a SHARD of a larger alignment must not
   //  choose by its own size — PAML_AMD_SHARD; paml_amd_comm_init checks)
   const bool small20 = n_states == 20 && !e->want_m20 && !(flags & (PAML_AMD_KEEP_PARTIALS | PAML_AMD_SHARD)) && n_patt <= 4096 && !e->env.valu20 && !e->env.no_coop;
   e->small20 = small20;
   const bool mfma20 = small20 || (n_states == 20 && !e->want_m20 && e->jit_enabled && !(flags & PAML_AMD_KEEP_PARTIALS) && n_tips <= 95 && !e->env.valu20);
   if (n_states == 4) e->kk = KK_VALU4;
   else if (n_states == 5) e->kk = KK_VALU5;
   else if (n_states == 20 && !mfma20) e->kk = KK_VALU20;
   else e->kk = KK_MFMA64;
   e->mfma_dma = n_tips <= MFMA_ZT && !e->env.force_gather;
   e->mfma_waves = e->mfma_dma ? DMA_WAVES : GATHER_WAVES;
   e->tile_patt = e->kk == KK_MFMA64 ? e->mfma_waves * 16 : 256;
   *out = e;
   return 0;
}

void paml_amd_destroy(paml_amd_engine *e)
{
   if (!e) return;
   (void)hipStreamSynchronize(e->stream);
   if (e->jit_job && e->jit_job->th.joinable()) e->jit_job->th.join();
   if (e->coop_job && e->coop_job->th.joinable()) e->coop_job->th.join();
   if (e->bjit_job && e->bjit_job->th.joinable()) e->bjit_job->th.join();
   for (hipStream_t s : e->sb)
      if (s) {
         (void)hipStreamSynchronize(s);
         (void)hipStreamDestroy(s);
      }
   if (e->s2) {
      (void)hipStreamSynchronize(e->s2);
      (void)hipStreamDestroy(e->s2);
      for (hipEvent_t ev : {e->ev_entry[0], e->ev_entry[1], e->ev_pmat}) if (ev) (void)hipEventDestroy(ev);
      for (hipEvent_t ev : e->ev_setread) if (ev) (void)hipEventDestroy(ev);
   }
   delete e;
}

const char *paml_amd_last_error(const paml_amd_engine *e) { return e ? e->err.c_str() : "null engine"; }

const char *paml_amd_kernel_name(const paml_amd_engine *e)
{
   if (!e) return "";
   switch (e->kk) {
   case KK_VALU4: return e->use_jit ? (e->fused ? (e->fused_mfma4 ? "mfma4_jit" : "valu4_fused_jit") : "valu4_jit") : "valu4";
   case KK_VALU5: return e->use_jit ? (e->fused ? "valu5_fused_jit" : "valu5_jit") : "valu5";
   case KK_VALU20: return e->use_jit ? (e->m20 ? "mfma4x20_jit" : "valu20_jit") : "valu20";
   default: return e->use_jit ? (e->jit_stage == 1 ? "mfma64_jit_quick" : "mfma64_jit") : (e->mfma_dma ? "mfma64_stream" : (e->coopj ? "mfma64_coopjit" : (e->coop ? "mfma64_coop" : "mfma64_gather")));
   }
}

int paml_amd_set_stream(paml_amd_engine *e, void *hip_stream)
{
   enter(e);
   if (!e) return PAML_AMD_EINVAL;
   (void)hipStreamSynchronize(e->stream);
   e->stream = (hipStream_t)hip_stream;
   return 0;
}

int paml_amd_set_tips(paml_amd_engine *e, const unsigned char *z, int cleandata, int n_codes, const int *n_chara,
                      const unsigned char *chara_map, const double *weights, const int *gene_off)
{
   enter(e);
   if (!e || !z || !weights) return fail(e, PAML_AMD_EINVAL, "set_tips: null argument");
   const int n = e->n;
   std::vector<int> nch;
   std::vector<unsigned char> cmap;
   if (cleandata || !n_chara || !chara_map) {
      if (!cleandata) return fail(e, PAML_AMD_EINVAL, "set_tips: cleandata=0 needs n_chara/chara_map");
      n_codes = n;
      nch.assign(n, 1);
      cmap.assign((size_t)n * n, 0);
      for (int i = 0; i < n; i++) cmap[(size_t)i * n] = (unsigned char)i;
   }
   else {
      if (n_codes < 1 || n_codes > 256) return fail(e, PAML_AMD_EINVAL, "set_tips: n_codes out of range");
      nch.assign(n_chara, n_chara + n_codes);
      cmap.assign(chara_map, chara_map + (size_t)n_codes * n);
      for (int c = 0; c < n_codes; c++) {
         if (nch[c] < 0 || nch[c] > n) return fail(e, PAML_AMD_EINVAL, "set_tips: n_chara out of range");
         for (int k = 0; k < nch[c]; k++)
            if (cmap[(size_t)c * n + k] >= n) return fail(e, PAML_AMD_EINVAL, "set_tips: chara_map state out of range");
      }
   }
   const size_t nz = (size_t)e->n_tips * e->n_patt;
   for (size_t i = 0; i < nz; i++)
      if (z[i] >= n_codes) return fail(e, PAML_AMD_EINVAL, "set_tips: character code >= n_codes");
   // More than 64 codes at 21 .. 64 states (61 sense codons + more than three ambiguous triplets, SetMapAmbiguity treesub.c:1218-1286): the
   // per-tree kernel's ring block holds a tip's rows of the codes 0 .. 63, and a lane whose code is beyond them adds up the rows of the
   // code's states itself (jit_tip_overflow) — as many LDS reads as the set has states.  Which ambiguous codes get the fast rows is the
   // engine's choice: the codes past the single states are renumbered by (cells that hold the code) x (states of its set), descending,
   // so that "missing" and whatever else is frequent sit below 64.  Invisible to the caller: codes only index the tip tables.
   std::vector<unsigned char> zperm;
   if (e->kk == KK_MFMA64 && n_codes > 64) {
      int plain = 0;
      while (plain < std::min(n, n_codes) && nch[plain] == 1 && cmap[(size_t)plain * n] == plain) plain++;
      std::vector<long> cnt(n_codes, 0);
      for (size_t i = 0; i < nz; i++) cnt[z[i]]++;
      std::vector<int> order;      // old code numbers, in their new order
      for (int c = 0; c < n_codes; c++) order.push_back(c);
      std::stable_sort(order.begin() + plain, order.end(), [&](int x, int y) { return cnt[x] * nch[x] > cnt[y] * nch[y]; });
      std::vector<unsigned char> new_of(n_codes);
      std::vector<int> nch2(n_codes);
      std::vector<unsigned char> cmap2((size_t)n_codes * n, 0);
      for (int c = 0; c < n_codes; c++) {
         new_of[order[c]] = (unsigned char)c;
         nch2[c] = nch[order[c]];
         memcpy(&cmap2[(size_t)c * n], &cmap[(size_t)order[c] * n], n);
      }
      nch.swap(nch2);
      cmap.swap(cmap2);
      zperm.resize(nz);
      for (size_t i = 0; i < nz; i++) zperm[i] = new_of[z[i]];
      z = zperm.data();
   }
   e->amb_ascending = true;
   for (int c = 64; c < n_codes; c++)
      for (int k = 1; k < nch[c]; k++)
         if (cmap[(size_t)c * n + k] <= cmap[(size_t)c * n + k - 1]) e->amb_ascending = false;
   e->cleandata = cleandata ? 1 : 0;
   e->n_codes = n_codes;
   e->plain_codes = 0;      // the leading codes that are one state each, the code itself (every code of clean data; the sense codons / amino acids / bases otherwise)
   while (e->plain_codes < std::min(n, n_codes) && nch[e->plain_codes] == 1 && cmap[(size_t)e->plain_codes * n] == e->plain_codes) e->plain_codes++;
   e->gene_off.assign(e->n_genes + 1, 0);
   if (gene_off) e->gene_off.assign(gene_off, gene_off + e->n_genes + 1);
   else {
      if (e->n_genes != 1) return fail(e, PAML_AMD_EINVAL, "set_tips: gene_off required when n_genes > 1");
      e->gene_off[1] = e->n_patt;
   }
   if (e->gene_off[0] != 0 || e->gene_off[e->n_genes] != e->n_patt)
      return fail(e, PAML_AMD_EINVAL, "set_tips: gene_off must span [0, n_patt]");
   for (int g = 0; g < e->n_genes; g++)
      if (e->gene_off[g + 1] < e->gene_off[g]) return fail(e, PAML_AMD_EINVAL, "set_tips: gene_off must not decrease");      // (a pattern shard may hold nothing of a gene)
   HIPCHK(upload(e->d_z, z, nz, e->stream));
   HIPCHK(upload(e->d_weights, weights, (size_t)e->n_patt, e->stream));
   HIPCHK(upload(e->d_n_chara, nch.data(), nch.size(), e->stream));
   HIPCHK(upload(e->d_chara_map, cmap.data(), cmap.size(), e->stream));
   HIPCHK(upload(e->d_gene_off, e->gene_off.data(), e->gene_off.size(), e->stream));
   if (e->kk == KK_VALU4 || e->kk == KK_VALU5 || e->kk == KK_VALU20) {      // pattern-major copy of the codes for the per-tree kernels
      e->zpm_words = ((e->n_tips + 3) / 4 + 3) / 4 * 4;
      HIPCHK(e->d_zpm.ensure((size_t)e->n_patt * e->zpm_words));
      launch_zpm(e->d_z.p, (long)e->n_patt, e->n_tips, e->n_patt, e->zpm_words, e->d_zpm.p, e->stream);
   }
   HIPCHK(hipStreamSynchronize(e->stream));
   e->tile_stash.built_for = 0;      // (the other tile size's tables described the previous data)
   int r = build_tiles(e);
   if (r) return r;
   {  // state sets of the character codes as bit masks (tip ends of a branch in the branch-local evaluation)
      std::vector<unsigned long long> mask(n_codes, 0);
      for (int c = 0; c < n_codes; c++)
         for (int k = 0; k < nch[c]; k++) mask[c] |= 1ull << cmap[(size_t)c * n + k];
      HIPCHK(upload(e->d_code_mask, mask.data(), mask.size(), e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
   }
   e->have_tips = true;
   e->partials_valid = false;
   e->bl.valid = false;
   return 0;
}

int paml_amd_set_tree(paml_amd_engine *e, int n_nodes, int root, const int *sons_ptr, const int *sons, const int *label,
                      const unsigned char *scale_node)
{
   enter(e);
   if (!e || !sons_ptr || !sons) return fail(e, PAML_AMD_EINVAL, "set_tree: null argument");
   if (n_nodes <= e->n_tips || root < 0 || root >= n_nodes) return fail(e, PAML_AMD_EINVAL, "set_tree: bad sizes");
   TreeDesc t;
   t.n_tips = e->n_tips; t.n_nodes = n_nodes; t.root = root;
   t.sons_ptr.assign(sons_ptr, sons_ptr + n_nodes + 1);
   if (t.sons_ptr[0] != 0) return fail(e, PAML_AMD_EINVAL, "set_tree: sons_ptr[0] != 0");
   for (int i = 0; i < n_nodes; i++)
      if (t.sons_ptr[i + 1] < t.sons_ptr[i]) return fail(e, PAML_AMD_EINVAL, "set_tree: sons_ptr not monotone");
   t.sons.assign(sons, sons + t.sons_ptr[n_nodes]);
   std::vector<int> seen(n_nodes, 0);
   for (int s : t.sons) {
      if (s < 0 || s >= n_nodes || s == root || seen[s]++) return fail(e, PAML_AMD_EINVAL, "set_tree: bad son index");
   }
   for (int i = 0; i < n_nodes; i++) {
      if (i != root && !seen[i]) return fail(e, PAML_AMD_EINVAL, "set_tree: node without father");
      if (i >= e->n_tips && t.is_leaf(i)) return fail(e, PAML_AMD_EINVAL, "set_tree: internal node without sons");
      if (i < e->n_tips && i != root && !t.is_leaf(i)) return fail(e, PAML_AMD_EINVAL, "set_tree: tip with sons");
   }
   t.label.assign(n_nodes, 0);
   if (label) t.label.assign(label, label + n_nodes);
   t.scale_node.assign(n_nodes, 0);
   t.scale_slot.assign(n_nodes, -1);
   if (scale_node) {
      for (int i = 0; i < n_nodes; i++) {
         t.scale_node[i] = scale_node[i] ? 1 : 0;
         if (t.scale_node[i]) t.scale_slot[i] = t.n_scale++;
      }
   }
   std::vector<unsigned char> leaf(n_nodes);
   for (int i = 0; i < n_nodes; i++) leaf[i] = t.is_leaf(i) ? 1 : 0;
   e->tree = std::move(t);
   HIPCHK(upload(e->d_label, e->tree.label.data(), e->tree.label.size(), e->stream));
   HIPCHK(upload(e->d_is_leaf, leaf.data(), leaf.size(), e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   e->have_tree = true;
   e->pres_valid = false;
   e->prog_valid = false;
   e->partials_valid = false;
   e->bl.valid = false;
   return 0;
}

int paml_amd_set_pi(paml_amd_engine *e, int n_pi, const double *pi)
{
   enter(e);
   if (!e || !pi || (n_pi != 1 && n_pi != e->n_genes)) return fail(e, PAML_AMD_EINVAL, "set_pi: bad arguments");
   const int n = e->n;
   std::vector<double> buf;
   if (e->kk == KK_MFMA64) {
      buf.assign((size_t)n_pi * 64, 0.0);
      for (int g = 0; g < n_pi; g++)
         for (int j = 0; j < n; j++) buf[(size_t)g * 64 + (j & 3) * 16 + (j >> 2)] = pi[(size_t)g * n + j];
   }
   else
      buf.assign(pi, pi + (size_t)n_pi * n);
   HIPCHK(upload(e->d_pi, buf.data(), buf.size(), e->stream));
   HIPCHK(upload(e->d_pi_plain, pi, (size_t)n_pi * n, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   e->n_pi = n_pi;
   e->have_pi = true;
   return 0;
}

static EigenHost *eigen_slot(paml_amd_engine *e, int set_id)
{
   if (!e || set_id < 0 || set_id > 4096) return nullptr;
   if ((size_t)set_id >= e->eigen.size()) e->eigen.resize(set_id + 1);
   e->eigen_dirty = true;
   e->pres_valid = false;
   e->partials_valid = false;
   e->bl.valid = false;
   return &e->eigen[set_id];
}

int paml_amd_set_eigen_uvroot(paml_amd_engine *e, int set_id, const double *U, const double *V, const double *Root)
{
   enter(e);
   EigenHost *h = eigen_slot(e, set_id);
   if (!h || !U || !V || !Root) return fail(e, PAML_AMD_EINVAL, "set_eigen_uvroot: bad arguments");
   const size_t n = e->n;
   HIPCHK(upload(h->U, U, n * n, e->stream));
   HIPCHK(upload(h->V, V, n * n, e->stream));
   HIPCHK(upload(h->Root, Root, n, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   h->kind = PAML_AMD_EIGEN_UVROOT;
   return 0;
}

// Q dense ([n_sets][n * n]) or, nnz > 0, as the elements (row[k] >= col[k]: lower triangle and diagonal) every matrix of the batch may have:
// vals[n_sets][nnz] — a codon matrix has 263 + 61 of 3 721, and the 30 KB per matrix from pageable memory were what a batch call's host time
// went to (tools/eigen_call_cost.py)
static int eigen_qrev_batch(paml_amd_engine *e, int n_sets, const int *set_ids, const double *Q, int nnz, const int *row, const int *col, const double *vals,
                            const double *pi, const double *scale)
{
   enter(e);
   if (!e || n_sets < 1 || !set_ids || !pi || (nnz > 0 ? !row || !col || !vals : !Q)) return fail(e, PAML_AMD_EINVAL, "set_eigen_qrev_batch: bad arguments");
   const size_t n = e->n;
   if (nnz > 0) {
      std::vector<char> seen(n * n, 0);
      for (int k = 0; k < nnz; k++) {
         if (row[k] < col[k] || col[k] < 0 || row[k] >= (int)n) return fail(e, PAML_AMD_EINVAL, "set_eigen_qrev_batch_sparse: an element outside the lower triangle");
         if (seen[(size_t)row[k] * n + col[k]]++) return fail(e, PAML_AMD_EINVAL, "set_eigen_qrev_batch_sparse: an element appears twice");
      }
   }
   int max_id = -1;
   for (int i = 0; i < n_sets; i++) {
      if (set_ids[i] < 0 || set_ids[i] > 4096) return fail(e, PAML_AMD_EINVAL, "set_eigen_qrev_batch: set id out of range");
      for (int j = 0; j < i; j++)
         if (set_ids[j] == set_ids[i]) return fail(e, PAML_AMD_EINVAL, "set_eigen_qrev_batch: a set id appears twice");
      max_id = std::max(max_id, set_ids[i]);
   }
   if (!eigen_slot(e, max_id)) return fail(e, PAML_AMD_EINVAL, "set_eigen_qrev_batch: bad arguments");      // (sizes the table once: the slots below do not move)
   std::vector<double *> ptr((size_t)5 * n_sets, nullptr);
   // Warm start: a decomposition starts from the eigenvectors of the NEAREST matrix any set of the engine was last decomposed for — its
   // own predecessor along a line search, the base point's sets for the perturbed points of a gradient (1e-6 away: two sweeps, where
   // the slot's own predecessor, a point of the previous iterate, takes four to six) — told by a signature of eight weighted sums over
   // the lower triangle's rows (the triangle the kernel reads).  A bad choice costs sweeps, never the result: every orthogonal start
   // with the same left-out states runs to the same stopping rule.  Every EIG_WARM_RUN-th decomposition of a chain starts cold again,
   // so that rounding in the accumulated rotations cannot build up.
   constexpr int EIG_WARM_RUN = 16;
   struct Pick { int src = -1; unsigned long long mask = 0; double sig[8]; };
   std::vector<Pick> pick(e->eigen_warm ? n_sets : 0);
   struct Cand { int idx; unsigned long long mask; double sig[8]; };
   std::vector<Cand> cands;      // the sets a start can come from, side by side (the search below is n_sets x their number)
   if (e->eigen_warm) {
      for (size_t sidx = 0; sidx < e->eigen.size(); sidx++) {
         const EigenHost &c = e->eigen[sidx];
         if (c.kind != PAML_AMD_EIGEN_UVROOT || c.warm_run < 0 || c.warm_run >= EIG_WARM_RUN - 1 || !c.Rt[c.rt_cur].p) continue;
         Cand cd;
         cd.idx = (int)sidx; cd.mask = c.live_mask;
         memcpy(cd.sig, c.sig, sizeof(cd.sig));
         cands.push_back(cd);
      }
      for (int i = 0; i < n_sets; i++) {
         Pick &pk = pick[i];
         const double *pii = pi + (size_t)i * n;
         for (size_t s = 0; s < n; s++)
            if (pii[s] > 1e-100) pk.mask |= 1ull << s;
         if (nnz > 0) {      // (the same sums over the elements handed over: row r's elements left of the diagonal)
            const double *vi = vals + (size_t)i * nnz;
            for (int k = 0; k < 8; k++) pk.sig[k] = 0;
            for (int x = 0; x < nnz; x++)
               if (row[x] != col[x])
                  for (int k = 0; k < 8; k++)
                     if ((size_t)row[x] == n - 1 - (size_t)k * (n / 9)) pk.sig[k] += vi[x] * (1.0 + 0.37 * (double)(((size_t)col[x] * 7 + k) % 5));
         }
         else {
            const double *Qi = Q + (size_t)i * n * n;
            for (int k = 0; k < 8; k++) {
               const size_t r = n - 1 - (size_t)k * (n / 9);
               double acc = 0;
               for (size_t j = 0; j < r; j++) acc += Qi[r * n + j] * (1.0 + 0.37 * (double)((j * 7 + k) % 5));
               pk.sig[k] = acc;
            }
         }
         double best = 0.25;      // (further than this: no better than a cold start)
         for (const Cand &c : cands) {
            if (c.mask != pk.mask) continue;
            double d = 0;
            for (int k = 0; k < 8 && d <= best; k++) d = std::max(d, std::fabs(c.sig[k] - pk.sig[k]) / (std::fabs(c.sig[k]) + std::fabs(pk.sig[k]) + 1e-300));
            if (d < best || (d == best && c.idx == set_ids[i])) { best = d; pk.src = c.idx; }
         }
      }
   }
   for (int i = 0; i < n_sets; i++) {
      EigenHost *h = eigen_slot(e, set_ids[i]);
      HIPCHK(h->U.ensure(n * n));
      HIPCHK(h->V.ensure(n * n));
      HIPCHK(h->Root.ensure(n));
      ptr[i] = h->U.p; ptr[n_sets + i] = h->V.p; ptr[2 * (size_t)n_sets + i] = h->Root.p;
      if (e->eigen_warm) {
         HIPCHK(h->Rt[0].ensure(64 * 64));
         HIPCHK(h->Rt[1].ensure(64 * 64));
      }
   }
   if (e->eigen_warm) {
      // (the sources are read as they were BEFORE this batch — rt_cur and warm_run of a set that is itself in the batch change below)
      std::vector<int> run(n_sets, 0);
      for (int i = 0; i < n_sets; i++) {
         const Pick &pk = pick[i];
         if (pk.src >= 0) {
            const EigenHost &c = e->eigen[pk.src];
            ptr[3 * (size_t)n_sets + i] = c.Rt[c.rt_cur].p;
            run[i] = c.warm_run + 1;
            e->n_eigen_warm++;
         }
      }
      for (int i = 0; i < n_sets; i++) {
         EigenHost *h = &e->eigen[set_ids[i]];
         ptr[4 * (size_t)n_sets + i] = h->Rt[h->rt_cur ^ 1].p;
      }
      for (int i = 0; i < n_sets; i++) {
         EigenHost *h = &e->eigen[set_ids[i]];
         h->rt_cur ^= 1;
         h->warm_run = run[i];
         h->live_mask = pick[i].mask;
         memcpy(h->sig, pick[i].sig, sizeof(h->sig));
         h->kind = PAML_AMD_EIGEN_UVROOT;
      }
   }
   else
      for (int i = 0; i < n_sets; i++) {
         EigenHost *h = &e->eigen[set_ids[i]];
         h->warm_run = -1;
         h->kind = PAML_AMD_EIGEN_UVROOT;
      }
   std::vector<double> ones;
   if (!scale) { ones.assign(n_sets, 1.0); scale = ones.data(); }
   // one upload for what the kernel reads besides the matrices — frequencies, scales, the table of pointers — and, sparse hand-over, one
   // for the values; the elements' positions only when they differ from the ones the device holds (an upload from pageable memory is
   // ~8 us of the host's time each, and a search makes a few hundred of these calls)
   const size_t npi = (size_t)n_sets * n, nsc = (size_t)n_sets, nptr = ptr.size();
   std::vector<double> pack(npi + nsc + nptr);
   memcpy(pack.data(), pi, npi * sizeof(double));
   memcpy(pack.data() + npi, scale, nsc * sizeof(double));
   static_assert(sizeof(double *) == sizeof(double), "the pointers ride in the same buffer");
   memcpy(pack.data() + npi + nsc, ptr.data(), nptr * sizeof(double *));
   HIPCHK(upload(e->d_eq_pi, pack.data(), pack.size(), e->stream));
   if (nnz > 0) {
      HIPCHK(upload(e->d_eq_q, vals, (size_t)n_sets * nnz, e->stream));
      std::vector<int> rc2((size_t)2 * nnz);
      for (int k = 0; k < nnz; k++) { rc2[2 * k] = row[k]; rc2[2 * k + 1] = col[k]; }
      if (rc2 != e->h_eq_rc) {
         HIPCHK(upload(e->d_eq_rc, rc2.data(), rc2.size(), e->stream));
         e->h_eq_rc.swap(rc2);
      }
   }
   else HIPCHK(upload(e->d_eq_q, Q, (size_t)n_sets * n * n, e->stream));
   HIPCHK(e->d_eq_sweeps.ensure(n_sets));
   if (!e->h_eig_fail) {
      HIPCHK(hipHostMalloc((void **)&e->h_eig_fail, 64, hipHostMallocDefault));
      *e->h_eig_fail = 0;
   }
   if (!e->eigen_attr_set) {
      for (const void *fn : {(const void *)eigen_qrev_kernel<0>, (const void *)eigen_qrev_kernel<20>, (const void *)eigen_qrev_kernel<60>, (const void *)eigen_qrev_kernel<62>})
         HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EIG_LDS_BYTES));
      e->eigen_attr_set = true;
   }
   EigenQrevArgs a{};
   double *const *dptr = (double *const *)(e->d_eq_pi.p + npi + nsc);
   a.n = (int)n; a.Q = e->d_eq_q.p; a.pi = e->d_eq_pi.p; a.scale = e->d_eq_pi.p + npi;
   a.nnz = nnz; a.rc = nnz > 0 ? e->d_eq_rc.p : nullptr;
   a.U = dptr; a.V = dptr + n_sets; a.Root = dptr + 2 * (size_t)n_sets; a.sweeps = e->d_eq_sweeps.p; a.fail = e->h_eig_fail;
   static const int sweep_limit = getenv("PAML_AMD_EIGEN_SWEEP_LIMIT") ? std::max(1, atoi(getenv("PAML_AMD_EIGEN_SWEEP_LIMIT"))) : 40;
   a.max_sweeps = sweep_limit;
   if (e->eigen_warm) { a.R0 = dptr + 3 * (size_t)n_sets; a.Rout = dptr + 4 * (size_t)n_sets; }
   // (the orders with a register form: R^T in a ninth wave's registers; PAML_AMD_EIGEN_LDS=1: the any-order form for them too)
   static const bool lds_form = getenv("PAML_AMD_EIGEN_LDS") != nullptr;
   const int N_even = ((int)n + 1) & ~1;
   if (N_even == 62 && !lds_form) hipLaunchKernelGGL(eigen_qrev_kernel<62>, dim3(n_sets), dim3(EIG_NT), EIG_LDS_BYTES, e->stream, a);
   else if (N_even == 60 && !lds_form) hipLaunchKernelGGL(eigen_qrev_kernel<60>, dim3(n_sets), dim3(EIG_NT), EIG_LDS_BYTES, e->stream, a);
   else if (N_even == 20 && !lds_form) hipLaunchKernelGGL(eigen_qrev_kernel<20>, dim3(n_sets), dim3(EIG_NT), EIG_LDS_BYTES, e->stream, a);
   else hipLaunchKernelGGL(eigen_qrev_kernel<0>, dim3(n_sets), dim3(EIG_NT), EIG_LDS_BYTES, e->stream, a);
   HIPCHK(hipGetLastError());
   // (the host arrays were pageable: the runtime has staged them on return; the evaluations that follow on the engine's stream see the sets)
   e->n_eigen_device += n_sets;
   e->eq_last_batch = n_sets;
   return 0;
}

int paml_amd_set_eigen_qrev_batch(paml_amd_engine *e, int n_sets, const int *set_ids, const double *Q, const double *pi, const double *scale)
{
   return eigen_qrev_batch(e, n_sets, set_ids, Q, 0, nullptr, nullptr, nullptr, pi, scale);
}

int paml_amd_set_eigen_qrev_batch_sparse(paml_amd_engine *e, int n_sets, const int *set_ids, int nnz, const int *row, const int *col, const double *vals,
                                         const double *pi, const double *scale)
{
   if (nnz < 1) return fail(e, PAML_AMD_EINVAL, "set_eigen_qrev_batch_sparse: no elements");
   return eigen_qrev_batch(e, n_sets, set_ids, nullptr, nnz, row, col, vals, pi, scale);
}

int paml_amd_set_eigen_warm_start(paml_amd_engine *e, int on, long *n_warm)
{
   if (!e) return PAML_AMD_EINVAL;
   if (n_warm) *n_warm = e->n_eigen_warm;
   if (on >= 0) e->eigen_warm = on != 0;
   return 0;
}

int paml_amd_get_eigen(paml_amd_engine *e, int set_id, double *U, double *V, double *Root)
{
   enter(e);
   if (!e || set_id < 0 || (size_t)set_id >= e->eigen.size() || e->eigen[set_id].kind != PAML_AMD_EIGEN_UVROOT)
      return fail(e, PAML_AMD_EINVAL, "get_eigen: not a U / V / Root set");
   const EigenHost &h = e->eigen[set_id];
   const size_t n = e->n;
   if (U) HIPCHK(hipMemcpyAsync(U, h.U.p, n * n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (V) HIPCHK(hipMemcpyAsync(V, h.V.p, n * n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (Root) HIPCHK(hipMemcpyAsync(Root, h.Root.p, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return 0;
}

int paml_amd_eigen_counters(paml_amd_engine *e, long *n_decomposed, int *sweeps_last_batch, int cap)
{
   enter(e);
   if (!e) return PAML_AMD_EINVAL;
   if (n_decomposed) *n_decomposed = e->n_eigen_device;
   int m = 0;
   if (sweeps_last_batch && e->d_eq_sweeps.p && e->eq_last_batch > 0) {
      m = std::min(cap, e->eq_last_batch);
      HIPCHK(hipMemcpyAsync(sweeps_last_batch, e->d_eq_sweeps.p, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
   }
   return m;
}

int paml_amd_set_eigen_cijk(paml_amd_engine *e, int set_id, int nR, const double *Cijk, const double *Root)
{
   enter(e);
   EigenHost *h = eigen_slot(e, set_id);
   if (!h || !Cijk || !Root || nR < 1 || nR > 64) return fail(e, PAML_AMD_EINVAL, "set_eigen_cijk: bad arguments");
   const size_t n = e->n;
   HIPCHK(upload(h->Cijk, Cijk, n * n * nR, e->stream));
   HIPCHK(upload(h->Root, Root, (size_t)nR, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   h->kind = PAML_AMD_EIGEN_CIJK;
   h->nR = nR;
   return 0;
}

int paml_amd_set_eigen_k80(paml_amd_engine *e, int set_id, double kappa)
{
   enter(e);
   if (e && e->n != 4) return fail(e, PAML_AMD_EINVAL, "set_eigen_k80: needs 4 states");
   EigenHost *h = eigen_slot(e, set_id);
   if (!h) return fail(e, PAML_AMD_EINVAL, "set_eigen_k80: bad arguments");
   h->kind = PAML_AMD_EIGEN_K80;
   h->kappa = kappa;
   return 0;
}

int paml_amd_set_eigen_jc69like(paml_amd_engine *e, int set_id)
{
   enter(e);
   EigenHost *h = eigen_slot(e, set_id);
   if (!h) return fail(e, PAML_AMD_EINVAL, "set_eigen_jc69like: bad arguments");
   h->kind = PAML_AMD_EIGEN_JC69LIKE;
   return 0;
}

int paml_amd_set_eigen_qmat(paml_amd_engine *e, int set_id, const double *Q)
{
   enter(e);
   EigenHost *h = eigen_slot(e, set_id);
   if (!h || !Q) return fail(e, PAML_AMD_EINVAL, "set_eigen_qmat: bad arguments");
   if (e->n > 8) return fail(e, PAML_AMD_EUNSUPPORTED, "set_eigen_qmat: at most 8 states");
   HIPCHK(upload(h->U, Q, (size_t)e->n * e->n, e->stream));      // the U slot carries Q
   HIPCHK(hipStreamSynchronize(e->stream));
   h->kind = PAML_AMD_EIGEN_QMAT;
   return 0;
}

int paml_amd_set_classes(paml_amd_engine *e, int mode, int K, const double *freqK, const double *rate, int n_labels,
                         const int *eigen_of, const double *qfactor)
{
   enter(e);
   if (!e || K < 1 || K > e->max_classes || n_labels < 1 || !eigen_of)
      return fail(e, PAML_AMD_EINVAL, "set_classes: bad arguments");
   if (mode != PAML_AMD_MODE_LFUN && mode != PAML_AMD_MODE_LFUNDG) return fail(e, PAML_AMD_EINVAL, "set_classes: bad mode");
   if (mode == PAML_AMD_MODE_LFUN && K != 1) return fail(e, PAML_AMD_EINVAL, "set_classes: lfun mode needs K = 1");
   if (e->have_tree)
      for (int lab : e->tree.label)
         if (lab < 0 || lab >= n_labels) return fail(e, PAML_AMD_EINVAL, "set_classes: tree label >= n_labels");
   std::vector<double> f(K, 1.0), r(K, 1.0), q((size_t)K * n_labels, 1.0);
   if (freqK) f.assign(freqK, freqK + K);
   if (rate) r.assign(rate, rate + K);
   if (qfactor) q.assign(qfactor, qfactor + (size_t)K * n_labels);
   HIPCHK(upload(e->d_freqK, f.data(), f.size(), e->stream));
   HIPCHK(upload(e->d_rate, r.data(), r.size(), e->stream));
   e->class_rate = r;
   e->rate_per_gene = false;
   HIPCHK(upload(e->d_qfactor, q.data(), q.size(), e->stream));
   HIPCHK(upload(e->d_eigen_of, eigen_of, (size_t)e->n_genes * K * n_labels, e->stream));
   e->h_eigen_of.assign(eigen_of, eigen_of + (size_t)e->n_genes * K * n_labels);
   e->h_qfactor = q;
   e->pres_valid = false;
   HIPCHK(hipStreamSynchronize(e->stream));
   e->mode = mode; e->K = K; e->n_labels = n_labels;
   e->have_classes = true;
   e->partials_valid = false;
   e->bl.valid = false;
   return 0;
}

int paml_amd_set_gene_class_rates(paml_amd_engine *e, const double *rate)
{
   if (!e || !e->have_classes) return fail(e, PAML_AMD_EINVAL, "set_gene_class_rates before set_classes");
   enter(e);
   if (!rate) {      // back to the class rates of set_classes (kept on the host for this)
      if (!e->rate_per_gene) return 0;
      HIPCHK(upload(e->d_rate, e->class_rate.data(), e->class_rate.size(), e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      e->rate_per_gene = false;
      e->pres_valid = false;
      e->partials_valid = false;
      e->bl.valid = false;
      return 0;
   }
   HIPCHK(upload(e->d_rate, rate, (size_t)e->n_genes * e->K, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   e->rate_per_gene = true;
   e->pres_valid = false;
   e->partials_valid = false;
   e->bl.valid = false;
   return 0;
}

int paml_amd_get_pmat(paml_amd_engine *e, int gene, int iclass, int node, double *P)
{
   enter(e);
   if (!e || !P || !e->d_rowmajor.p) return fail(e, PAML_AMD_EINVAL, "get_pmat: nothing evaluated yet");
   if (!e->pmat_valid)
      return fail(e, PAML_AMD_EINVAL, "get_pmat: the P(t) buffers hold the re-rooted matrices of eval_branch / node_posterior; run an evaluation first");
   if (gene < 0 || gene >= e->n_genes || iclass < 0 || iclass >= e->K || node < 0 || node >= e->tree.n_nodes ||
       node == e->tree.root)
      return fail(e, PAML_AMD_EINVAL, "get_pmat: index out of range");
   const size_t nn2 = (size_t)e->n * e->n;
   // (after a batched evaluation the kernels saw K x B classes per gene: element 0's matrices)
   const size_t slot = ((size_t)gene * e->K * std::max(1, e->pmat_B) + iclass) * e->tree.n_nodes + node;
   if (e->rowmajor_valid) {
      const double *src = e->d_rowmajor.p + slot * nn2;
      HIPCHK(hipMemcpyAsync(P, src, nn2 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      return 0;
   }
   // The matrix-core P(t) kernel writes only what the pruning kernels read (kernels_pmat.h): P is put together from that —
   // an internal branch's A-operand order copy, a tip branch's column table (the rows of the plain states: codes 0 .. n - 1).
   const int n = e->n;
   if (!e->tree.is_leaf(node)) {
      std::vector<double> f(4096);
      HIPCHK(hipMemcpyAsync(f.data(), e->d_pint.p + slot * 4096, 4096 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      for (int i = 0; i < n; i++)
         for (int j = 0; j < n; j++) {
            const int jb = i >> 4, kb = j >> 2, lane = (j & 3) * 16 + (i & 15);
            P[(size_t)i * n + j] = f[(((kb >> 1) * 4 + jb) * 64 + lane) * 2 + (kb & 1)];
         }
      return 0;
   }
   // (... which holds for every code table the reference builds — the states come first — but is the caller's choice at this boundary)
   if (e->plain_codes < n)
      return fail(e, PAML_AMD_EUNSUPPORTED, "get_pmat: a tip branch's matrix is rebuilt from the tip's column table, which needs the codes 0 .. n-1 to be the "
                                            "single states (set_tips); PAML_AMD_PMAT_ROWMAJOR=1 keeps the row-major copies");
   const size_t tw = tip_words(e);
   std::vector<double> tab(tw);
   HIPCHK(hipMemcpyAsync(tab.data(), e->d_ptip.p + slot * tw, tw * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   for (int jj = 0; jj < n; jj++)
      for (int c = 0; c < n; c++) {
         const int q = jj & 3, m = jj >> 2, row = c * 4 + q;
         const int w = q * 16 + ((((m >> 1) ^ TIP_SWZ(row)) & 7) << 1) + (m & 1);
         P[(size_t)jj * n + c] = tab[(size_t)c * 64 + w];
      }
   return 0;
}

int paml_amd_get_partials(paml_amd_engine *e, int node, int iclass, double *conP)
{
   enter(e);
   if (!e || !conP) return fail(e, PAML_AMD_EINVAL, "get_partials: null argument");
   if (!(e->flags & PAML_AMD_KEEP_PARTIALS) || !e->partials_valid)
      return fail(e, PAML_AMD_EINVAL, "get_partials: needs PAML_AMD_KEEP_PARTIALS and a completed evaluation");
   if (node < e->n_tips || node >= e->tree.n_nodes || iclass < 0 || iclass >= e->K)
      return fail(e, PAML_AMD_EINVAL, "get_partials: index out of range");
   const int n = e->n, n_int = e->tree.n_nodes - e->n_tips;
   if (e->kk != KK_MFMA64) {
      const double *src = e->d_partials.p + ((size_t)iclass * n_int + (node - e->n_tips)) * e->n_patt * n;
      HIPCHK(hipMemcpyAsync(conP, src, (size_t)e->n_patt * n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      return 0;
   }
   const int MW = GATHER_WAVES;      // (the resident partials' one layout: groups in the order of the 64-pattern tile table)
   const size_t groups = (size_t)e->part_groups();
   std::vector<double> raw(groups * 1024);
   const double *src = e->d_partials.p + ((size_t)iclass * n_int + (node - e->n_tips)) * groups * 1024;
   HIPCHK(hipMemcpyAsync(raw.data(), src, raw.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   // native [group] x part_index(m, lane) -> [h][state]; lane = (state & 3) * 16 + (h & 15), m = state >> 2
   std::vector<int2> tiles;
   for (int g = 0; g < e->n_genes; g++)
      for (int h = e->gene_off[g]; h < e->gene_off[g + 1]; h += MW * 16) tiles.push_back(make_int2(g, h));
   for (size_t t = 0; t < tiles.size(); t++) {
      const int hend = e->gene_off[tiles[t].x + 1];
      for (int w = 0; w < MW; w++)
         for (int hl = 0; hl < 16; hl++) {
            const int h = tiles[t].y + w * 16 + hl;
            if (h >= hend) continue;
            const double *grp = raw.data() + (t * MW + w) * 1024;
            for (int j = 0; j < n; j++) conP[(size_t)h * n + j] = grp[part_index(j >> 2, (j & 3) * 16 + hl)];
         }
   }
   return 0;
}

int paml_amd_get_scale(paml_amd_engine *e, int node, int iclass, double *scale)
{
   enter(e);
   if (!e || !scale) return fail(e, PAML_AMD_EINVAL, "get_scale: null argument");
   if (!(e->flags & PAML_AMD_KEEP_PARTIALS) || !e->partials_valid)
      return fail(e, PAML_AMD_EINVAL, "get_scale: needs PAML_AMD_KEEP_PARTIALS and a completed evaluation");
   if (node < 0 || node >= e->tree.n_nodes || iclass < 0 || iclass >= e->K || e->tree.scale_slot[node] < 0)
      return fail(e, PAML_AMD_EINVAL, "get_scale: not a scaling node");
   const double *src = e->d_scalef.p + ((size_t)iclass * e->tree.n_scale + e->tree.scale_slot[node]) * e->n_patt;
   HIPCHK(hipMemcpyAsync(scale, src, (size_t)e->n_patt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return 0;
}

int paml_amd_profile(paml_amd_engine *e, int enable)
{
   if (!e) return PAML_AMD_EINVAL;
   enter(e);      // (a run of eval_device calls ends here: the per-kernel events are taken on the engine's stream, one evaluation at a time)
   e->profiling = enable != 0;
   return 0;
}

int paml_amd_profile_read(paml_amd_engine *e, double *ms_pmat, double *ms_prune, double *ms_reduce, long *n_evals)
{
   if (!e) return PAML_AMD_EINVAL;
   HIPCHK(hipStreamSynchronize(e->stream));
   double acc[3] = {0, 0, 0};
   for (size_t i = 0; i + 5 < e->ev_used.size(); i += 6)
      for (int k = 0; k < 3; k++) {
         float ms = 0;
         if (hipEventElapsedTime(&ms, e->ev_used[i + 2 * k], e->ev_used[i + 2 * k + 1]) == hipSuccess) acc[k] += ms;
      }
   for (auto ev : e->ev_used) e->ev_pool.push_back(ev);
   e->ev_used.clear();
   if (ms_pmat) *ms_pmat = acc[0];
   if (ms_prune) *ms_prune = acc[1];
   if (ms_reduce) *ms_reduce = acc[2];
   if (n_evals) *n_evals = e->prof_evals;
   e->prof_evals = 0;
   return 0;
}

int paml_amd_counters(const paml_amd_engine *e, long *n_eval, long *n_pmat)
{
   if (!e) return PAML_AMD_EINVAL;
   if (n_eval) *n_eval = e->n_eval;
   if (n_pmat) *n_pmat = e->n_pmat;
   return 0;
}

}  // extern "C"
