// engine_branch.hip — the tree seen from another node: branch-local lnL(t), dlnL/dt, d2lnL/dt2 on resident partials
// (paml_amd_eval_branch: lfuntdd / minbranches, treesub.c:7826-8541) and the marginal posteriors at a node (paml_amd_node_posterior).
// Built for gfx950 only (one of the translation units of libpaml_amd.so, see engine_state.h).
#include "engine_state.h"
#include "kernels_branch.h"

namespace paml_amd {
namespace {

// the contraction kernel's instantiations: [how A is obtained][B is a tip]
typedef void (*beig_fn)(BranchEigArgs);
#define BEIG_ROW(NS, S0, S1) {{branch_eig_kernel<NS, S0, S1, false, false>, branch_eig_kernel<NS, S0, S1, false, true>}, \
                              {branch_eig_kernel<NS, S0, S1, true, false>, branch_eig_kernel<NS, S0, S1, true, true>}}
beig_fn const beig_kernels[6][2][2] = {      // [how A is obtained][B is a tip][61 states]
   BEIG_ROW(0, false, false),      // A resident
   BEIG_ROW(1, true, false),       // one son, internal
   BEIG_ROW(1, false, false),      // one son, a tip
   BEIG_ROW(2, true, true),        // two internal sons
   BEIG_ROW(2, true, false),       // an internal son and a tip
   BEIG_ROW(2, false, false),      // two tips
};
#undef BEIG_ROW

// Run `prog` with the full-featured kernels (gather / valu) over all patterns and classes, reading the P(t) buffers
// of the last pmat launch; OP_EXPORT writes to export_buf.  Used by the branch-local evaluation.
int run_prune_full(paml_amd_engine *e, const Program &prog, double *export_buf, double *export_scale)
{
   const int nn = e->tree.n_nodes, K = e->K;
   HIPCHK(upload(e->d_ops_tmp, prog.ops.data(), prog.ops.size(), e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   const int waves = GATHER_WAVES;
   const int n_blocks = e->n_tiles_full * K;
   int overflow = 0;
   if (e->kk == KK_MFMA64 && prog.max_stack > MFMA_RS) {
      overflow = prog.max_stack - MFMA_RS;
      HIPCHK(e->d_stack.ensure((size_t)n_blocks * overflow * waves * 1024));
   }
   const int maxd = e->kk == KK_VALU20 ? VALU_MAXD_20 : VALU_MAXD_SMALL;
   if (e->kk != KK_MFMA64 && prog.max_stack > maxd)
      return fail(e, PAML_AMD_EUNSUPPORTED, "tree needs a deeper partial stack than this kernel provides");
   PruneArgs pr{};
   pr.ops = e->d_ops_tmp.p; pr.z = e->d_z.p; pr.z_stride = e->n_patt; pr.tiles = e->d_tiles_full.p; pr.n_tiles = e->n_tiles_full;
   pr.gene_off = e->d_gene_off.p; pr.weights = e->d_weights.p;
   pr.n = e->n; pr.n_tips = e->n_tips; pr.n_nodes = nn; pr.K = K; pr.n_genes = e->n_genes; pr.n_codes = e->n_codes;
   pr.cleandata = e->cleandata; pr.n_pi = e->n_pi; pr.mode = e->mode; pr.n_scale = 0; pr.keep = 0; pr.n_patt = e->n_patt;
   pr.pi = e->d_pi.p; pr.pint = e->kk == KK_MFMA64 ? e->d_pint.p : e->d_rowmajor.p; pr.ptip = e->d_ptip.p;
   pr.fhK = e->d_fhK.p; pr.partials = nullptr; pr.scalef = nullptr; pr.stack_scratch = e->d_stack.p;
   pr.stack_overflow_slots = overflow; pr.first_matmul = prog.first_matmul; pr.n_int = nn - e->n_tips;
   pr.first_tip = prog.first_tip; pr.tip_words = (long)tip_words(e); pr.export_buf = export_buf; pr.export_scale = export_scale;
   launch_prune_full(e, prog.max_stack, n_blocks, pr, e->stream);
   HIPCHK(hipGetLastError());
   return 0;
}

// Branch-local evaluation and node posteriors look at the tree from another node: build the tree rooted at `new_root`
// (along the path new_root -> old root every node loses the son it came from and gains its father; the edge data —
// length, label — of node p moves to its father, now a son of p; cut_son >= 0: that son of new_root and its subtree are
// left out), send the re-oriented branch lengths / labels, and compute P(t) for every edge with one batched launch.
// What ReRootTree (treespace.c:236) + updateconP (treesub.c:7982) do on the host in the reference.
int rerooted_pmat(paml_amd_engine *e, int new_root, int cut_son, const double *branch, const double *gene_rate, TreeDesc *out)
{
   const TreeDesc &T = e->tree;
   const int nn = T.n_nodes, n = e->n, K = e->K, G = e->n_genes, psets = G * K;
   std::vector<int> father(nn, -1);
   for (int i = 0; i < nn; i++)
      for (int j = T.sons_ptr[i]; j < T.sons_ptr[i + 1]; j++) father[T.sons[j]] = i;
   std::vector<std::vector<int>> sons(nn);
   for (int i = 0; i < nn; i++) sons[i].assign(T.sons.begin() + T.sons_ptr[i], T.sons.begin() + T.sons_ptr[i + 1]);
   std::vector<double> br(branch, branch + nn);
   std::vector<int> lab(T.label);
   for (int p = new_root, prev = cut_son; p >= 0; prev = p, p = father[p]) {
      auto &s = sons[p];
      if (prev >= 0) s.erase(std::find(s.begin(), s.end(), prev));
      if (father[p] >= 0) {
         s.push_back(father[p]);
         br[father[p]] = branch[p];
         lab[father[p]] = T.label[p];
      }
   }
   TreeDesc t;
   t.n_tips = T.n_tips; t.n_nodes = nn; t.root = new_root;
   t.sons_ptr.assign(nn + 1, 0);
   for (int i = 0; i < nn; i++) t.sons_ptr[i + 1] = t.sons_ptr[i] + (int)sons[i].size();
   for (int i = 0; i < nn; i++) t.sons.insert(t.sons.end(), sons[i].begin(), sons[i].end());
   t.label = lab;
   // the nodes SetNodeScale marked keep rescaling their partial, whichever subtree it now stands for; the factors
   // travel with the exported partials
   t.scale_node.assign(nn, 0);
   t.scale_slot.assign(nn, -1);
   if (T.n_scale > 0)
      for (int i = 0; i < nn; i++)
         if (T.scale_node[i] && !t.is_leaf(i)) { t.scale_node[i] = 1; t.scale_slot[i] = t.n_scale++; }
   *out = t;

   std::vector<double> gr(G, 1.0);
   if (gene_rate) gr.assign(gene_rate, gene_rate + G);
   HIPCHK(upload(e->d_branch, br.data(), br.size(), e->stream));
   HIPCHK(upload(e->d_gene_rate, gr.data(), gr.size(), e->stream));
   e->bl_gr_sent = false;
   HIPCHK(upload(e->d_label_eff, lab.data(), lab.size(), e->stream));
   if (e->eigen_dirty) {
      std::vector<EigenDev> tab(e->eigen.size());
      for (size_t i = 0; i < e->eigen.size(); i++) {
         const EigenHost &h = e->eigen[i];
         tab[i] = EigenDev{h.kind, h.nR, h.kappa, h.U.p, h.V.p, h.Root.p, h.Cijk.p};
      }
      HIPCHK(upload(e->d_eigen, tab.data(), tab.size(), e->stream));
      e->eigen_dirty = false;
   }
   HIPCHK(hipStreamSynchronize(e->stream));
   HIPCHK(e->d_rowmajor.ensure((size_t)psets * nn * n * n));
   if (e->kk == KK_MFMA64) HIPCHK(e->d_pint.ensure((size_t)psets * nn * 4096));
   HIPCHK(e->d_ptip.ensure((size_t)psets * nn * tip_words(e)));
   HIPCHK(e->d_fhK.ensure((size_t)K * e->n_patt));
   PmatArgs pa{};
   pa.n = n; pa.n_nodes = nn; pa.root = new_root; pa.K = K; pa.n_genes = G; pa.n_labels = e->n_labels;
   pa.n_codes = e->n_codes; pa.layout = e->kk == KK_MFMA64 ? 1 : 0;
   pa.label = e->d_label_eff.p; pa.is_leaf = e->d_is_leaf.p; pa.branch = e->d_branch.p; pa.rate = e->d_rate.p;
   pa.gene_rate = e->d_gene_rate.p; pa.eigen_of = e->d_eigen_of.p; pa.qfactor = e->d_qfactor.p;
   pa.eigen = e->d_eigen.p; pa.n_chara = e->d_n_chara.p; pa.chara_map = e->d_chara_map.p; pa.plain_codes = e->plain_codes;
   pa.rowmajor = e->d_rowmajor.p; pa.pint = e->d_pint.p; pa.ptip = e->d_ptip.p; pa.tip_words = (long)tip_words(e);
   pa.B = 1; pa.rate_gs = e->rate_per_gene ? K : 0;
   {
      InlineVec iv;
      iv.n_branch = iv.n_rate = 0;
      launch_pmat(pa, iv, nn, psets, false, e->stream, pmat_on_matrix_cores(e, pa));
   }
   e->n_pmat += (long)psets * (nn - 1);
   e->prog_valid = false;      // d_branch / P buffers now hold the re-rooted edge data: the next eval rebuilds
   e->partials_valid = false;
   e->pmat_valid = false;
   return 0;
}

}  // namespace
}  // namespace paml_amd

extern "C" {

int paml_amd_eval_branch(paml_amd_engine *e, int node_b, int n_t, const double *t, const double *branch,
                         const double *gene_rate, double *lnL, double *dlnL, double *ddlnL)
{
   enter(e);
   if (!e || !t || !branch || !lnL || !dlnL || !ddlnL || n_t < 1 || n_t > 64)
      return fail(e, PAML_AMD_EINVAL, "eval_branch: bad arguments");
   if (!(e->have_tips && e->have_tree && e->have_pi && e->have_classes) || e->eigen.empty())
      return fail(e, PAML_AMD_EINVAL, "eval_branch before set_tips/set_tree/set_pi/set_classes/set_eigen");
   const TreeDesc &T = e->tree;
   const int nn = T.n_nodes, n = e->n, K = e->K, G = e->n_genes, psets = G * K, n_int = nn - e->n_tips;
   if (node_b < 0 || node_b >= nn || node_b == T.root) return fail(e, PAML_AMD_EINVAL, "eval_branch: node has no branch");
   for (size_t i = 0; i < e->eigen.size(); i++)
      if (e->eigen[i].kind == PAML_AMD_EIGEN_QMAT) return fail(e, PAML_AMD_EUNSUPPORTED, "eval_branch: not for rate-matrix (UNREST) sets");
   const bool mfma = e->kk == KK_MFMA64;
   std::vector<int> father(nn, -1);
   std::vector<std::vector<int>> nbr(nn);
   for (int i = 0; i < nn; i++)
      for (int j = T.sons_ptr[i]; j < T.sons_ptr[i + 1]; j++) {
         father[T.sons[j]] = i;
         nbr[i].push_back(T.sons[j]);
         nbr[T.sons[j]].push_back(i);
      }
   auto edge_id = [&](int u, int v) { return father[u] == v ? u : v; };      // an edge is named by its lower node in the tree as set
   // the two ends of the branch; the end that may be a tip is "b" (the contraction is symmetric for reversible models:
   // pi_i P_ij = pi_j P_ji)
   int A = father[node_b], Bn = node_b;
   if (T.is_leaf(A)) std::swap(A, Bn);
   if (T.is_leaf(A)) return fail(e, PAML_AMD_EUNSUPPORTED, "eval_branch: a branch between two tips");
   const bool b_tip = T.is_leaf(Bn);

   // ---- the message cache: what updateconP (treesub.c:7982) + com.oldconP (treespace.c:250) save the reference --------
   // Every internal node v keeps one partial M[v]: the likelihood of everything on v's side of the edge (v, up[v]).  With
   // all up[] pointing towards the branch being worked on, M[A] and M[B] are the two partials across it.  Moving to another
   // branch re-orients only the nodes on the path between the two branches; a changed branch length invalidates only the
   // partials that look across it.  Nothing else is recomputed.
   paml_amd_engine::BranchCache &bc = e->bl;
   // The cache's bookkeeping (which partials, coefficients and operand tables are current) is updated where the kernels that fill them are
   // queued; should anything after that fail — a launch, the exchange step, the final synchronisation — the call returns its error and the
   // whole cache is dropped, so that the next call cannot take a "hit" on buffers that were never written.
   struct DropCacheOnError {
      paml_amd_engine::BranchCache &c;
      bool ok = false;
      ~DropCacheOnError() { if (!ok) c.valid = false; }
   } cache_guard{bc};
   const size_t words = mfma ? (size_t)K * n_int * e->n_tiles_full * GATHER_WAVES * 1024 : (size_t)K * n_int * e->n_patt * n;
   if (words + (mfma ? 8 * 1024 : 0) > e->d_bl_partials.cap) { HIPCHK(e->d_bl_partials.ensure(words + (mfma ? 8 * 1024 : 0))); bc.valid = false; }      // (+ PruneArgs::part_dump)
   const bool scaled = T.n_scale > 0;
   if (scaled && (size_t)K * T.n_scale * e->n_patt > e->d_bl_scalef.cap) { HIPCHK(e->d_bl_scalef.ensure((size_t)K * T.n_scale * e->n_patt)); bc.valid = false; }
   std::vector<double> gr(G, 1.0);
   if (gene_rate) gr.assign(gene_rate, gene_rate + G);
   if (!bc.valid || (int)bc.up.size() != nn || bc.K != K || bc.gr != gr) {
      bc.up.assign(nn, -2); bc.ok.assign(nn, 0); bc.br.assign(nn, -1.0); bc.gr = gr; bc.K = K;
      bc.valid = true;
      bc.coef_ok = false;
      bc.frag_ok.clear();
      e->bl_gr_sent = false;
   }
   {  // branch lengths that changed since the partials were formed
      std::vector<int> changed;
      for (int x = 0; x < nn; x++)
         if (x != T.root && branch[x] != bc.br[x]) { changed.push_back(x); bc.br[x] = branch[x]; }
      if (!changed.empty()) {
         std::vector<char> in(nn);
         std::vector<int> stack;
         for (int v = e->n_tips; v < nn; v++) {
            if (!bc.ok[v]) continue;
            std::fill(in.begin(), in.end(), 0);      // v's side of the edge (v, up[v])
            stack.assign(1, v);
            in[v] = 1;
            while (!stack.empty()) {
               const int u = stack.back();
               stack.pop_back();
               for (int w : nbr[u])
                  if (!in[w] && !(u == v && w == bc.up[v])) { in[w] = 1; stack.push_back(w); }
            }
            for (int x : changed)
               if (in[x] && in[father[x]]) { bc.ok[v] = 0; break; }
         }
      }
   }
   // orientation towards the branch
   std::vector<int> up(nn, -1);
   {
      std::vector<int> queue;
      up[A] = Bn; up[Bn] = A;
      queue.push_back(A); queue.push_back(Bn);
      for (size_t qi = 0; qi < queue.size(); qi++) {
         const int u = queue[qi];
         for (int w : nbr[u])
            if (w != up[u] && up[w] < 0) { up[w] = u; queue.push_back(w); }
      }
   }
   std::vector<unsigned char> clean(nn, 0);
   bool any_dirty = false;
   for (int v = e->n_tips; v < nn; v++) {
      clean[v] = bc.ok[v] && bc.up[v] == up[v];
      any_dirty = any_dirty || !clean[v];
   }
   // the tree seen from the branch: sons = neighbours other than up[]; the edge data of (v, up[v]) sits at index v
   TreeDesc tr;
   tr.n_tips = T.n_tips; tr.n_nodes = nn; tr.root = A;
   tr.sons_ptr.assign(nn + 1, 0);
   std::vector<double> br_eff(nn, 0.0);
   std::vector<int> lab_eff(nn, 0);
   for (int v = 0; v < nn; v++) {
      for (int w : nbr[v])
         if (w != up[v]) tr.sons.push_back(w);
      tr.sons_ptr[v + 1] = (int)tr.sons.size();
      if (v != A && v != Bn) { const int x = edge_id(v, up[v]); br_eff[v] = branch[x]; lab_eff[v] = T.label[x]; }
   }
   tr.label = lab_eff;
   tr.scale_node.assign(nn, 0);
   tr.scale_slot.assign(nn, -1);
   if (scaled)
      for (int i = 0; i < nn; i++)
         if (T.scale_node[i] && !tr.is_leaf(i)) { tr.scale_node[i] = 1; tr.scale_slot[i] = T.scale_slot[i]; tr.n_scale = T.n_scale; }

   hipStream_t st = e->stream;
   if (int rc = eigen_refs_ok(e, e->h_eigen_of.data(), e->h_eigen_of.size(), "eval_branch")) return rc;
   if (e->eigen_dirty) {
      std::vector<EigenDev> tab;
      if (int rc = eigen_table(e, tab)) return rc;
      HIPCHK(upload(e->d_eigen, tab.data(), tab.size(), st));
      e->eigen_dirty = false;
   }
   // (d_gene_rate is shared with the ordinary evaluation, which rewrites it: sent again unless the last writer was this function with the same rates)
   if (!e->bl_gr_sent) { HIPCHK(upload(e->d_gene_rate, gr.data(), gr.size(), st)); }
   e->bl_gr_sent = true;
   // ---- the eigen-basis form (kernels_branch.h): matrix-core engines with one gene and (U, V, Root) eigen systems ------------------
   bool eig = mfma && G == 1 && e->n_pi == 1 && !e->env.no_branch_eig && (size_t)(K * BEIG_NT * 192 + 8 * 3 * BEIG_NT) * 8 <= 150 * 1024;
   for (const EigenHost &h : e->eigen) eig = eig && (h.kind == PAML_AMD_EIGEN_UVROOT || h.kind < 0);
   if (eig) {
      const int n_groups = e->n_tiles_full * GATHER_WAVES, n_out = 3 * n_t;
      const int chunk = e->chunk, cg = chunk / 16, nb_local = (e->n_patt + chunk - 1) / chunk, nbg = e->nb_global;
      const bool hit = bc.coef_ok && bc.coef_node == node_b && clean[A] && (b_tip || clean[Bn]) && !e->env.no_coef_cache;
      if (!e->beig_attr_set) {
         for (auto &row : beig_kernels)
            for (auto &r2 : row)
               for (auto fn : r2) HIPCHK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
         HIPCHK(hipFuncSetAttribute((const void *)branch_poly_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
         e->beig_attr_set = true;
      }
      HIPCHK(e->d_bl_coef.ensure((size_t)K * n_groups * 1024));
      const int NL = e->n_labels, lab_b = T.label[node_b];
      if ((size_t)NL * K * 2 * 4096 > e->d_bl_efrag.cap || (size_t)NL * K * e->n_codes * 64 > e->d_bl_ztab.cap || (size_t)NL * K * 128 > e->d_bl_ecol.cap ||
          (int)bc.frag_ok.size() != NL) {
         HIPCHK(e->d_bl_ecol.ensure((size_t)NL * K * 128));
         HIPCHK(e->d_bl_efrag.ensure((size_t)NL * K * 2 * 4096));
         HIPCHK(e->d_bl_ztab.ensure((size_t)NL * K * e->n_codes * 64));
         bc.frag_ok.assign(NL, 0);
      }
      if (lab_b < 0 || lab_b >= NL) return fail(e, PAML_AMD_EINVAL, "eval_branch: branch label out of range");
      double *const efrag = e->d_bl_efrag.p + (size_t)lab_b * K * 2 * 4096, *const ztab = e->d_bl_ztab.p + (size_t)lab_b * K * e->n_codes * 64;
      HIPCHK(e->d_bl_etab.ensure((size_t)K * n_t * 192));
      const int nrows = nbg * 8;      // a row per wave and chunk (kernels_branch.h): the sums of an eighth of a reduction chunk
      HIPCHK(e->d_bpartial.ensure((size_t)nrows * n_out));
      HIPCHK(e->d_bout.ensure((size_t)n_out));
      HIPCHK(e->d_tt.ensure(n_t));
      e->bpart_rows = nrows; e->bpart_cols = n_out; e->bpart_colmajor = true;
      if (nbg != nb_local || chunk * nb_local != e->n_patt) HIPCHK(hipMemsetAsync(e->d_bpartial.p, 0, (size_t)nrows * n_out * sizeof(double), st));      // (the other ranks' rows; rows past the last pattern)
      int n_sons = 0, son[2] = {-1, -1};
      Program prog;
      bool run_pmat = false;
      if (!hit && any_dirty) {
         // A itself is formed inside the contraction kernel when it has one or two sons in the tree seen from the branch (what
         // changes when minbranches moves on to a neighbouring branch); everything else that is dirty goes through the interpreter
         std::vector<int> roots;
         if (!clean[A]) {
            const int ns = tr.sons_ptr[A + 1] - tr.sons_ptr[A];
            if (!scaled && (ns == 1 || ns == 2)) {
               for (int j = tr.sons_ptr[A]; j < tr.sons_ptr[A + 1]; j++) son[n_sons++] = tr.sons[j];
               if (n_sons == 2 && T.is_leaf(son[0]) && !T.is_leaf(son[1])) std::swap(son[0], son[1]);      // (an internal son first: its product initialises the partial)
               for (int j = 0; j < n_sons; j++)
                  if (!T.is_leaf(son[j]) && !clean[son[j]]) roots.push_back(son[j]);
            }
            else roots.push_back(A);
         }
         if (!b_tip && !clean[Bn]) roots.push_back(Bn);
         for (int rt : roots) {
            tr.root = rt;
            Program ps = build_program(tr, true, clean.data());
            for (const Op &o : ps.ops)
               if (o.code != OP_ROOT && o.code != OP_END) prog.ops.push_back(o);
            prog.max_stack = std::max(prog.max_stack, ps.max_stack);
         }
         tr.root = A;
         run_pmat = true;
      }
      const bool run_prog = !prog.ops.empty();
      if (run_prog) {
         prog.ops.push_back({OP_END, 0, 0, -1});
         int next = -1;
         for (int i = (int)prog.ops.size() - 1; i >= 0; i--)
            if (prog.ops[i].code == OP_MATMUL || prog.ops[i].code == OP_MATMUL_POP) { prog.ops[i].c = next; next = prog.ops[i].a; }
         prog.first_matmul = next;
      }
      // the call's small inputs: one pinned arena, asynchronous copies
      HIPCHK(e->stage.begin((size_t)n_t * 8 + (run_pmat ? (size_t)nn * 12 : 0) + (run_prog ? prog.ops.size() * sizeof(Op) : 0) + 256));
      {
         const double *ht = e->stage.put(t, (size_t)n_t);
         HIPCHK(hipMemcpyAsync(e->d_tt.p, ht, (size_t)n_t * 8, hipMemcpyHostToDevice, st));
      }
      if (run_pmat) {
         HIPCHK(e->d_label_eff.ensure(nn));
         HIPCHK(e->d_branch.ensure(nn));
         const int *hl = e->stage.put(lab_eff.data(), (size_t)nn);
         HIPCHK(hipMemcpyAsync(e->d_label_eff.p, hl, (size_t)nn * 4, hipMemcpyHostToDevice, st));
         const double *hb = e->stage.put(br_eff.data(), (size_t)nn);
         HIPCHK(hipMemcpyAsync(e->d_branch.p, hb, (size_t)nn * 8, hipMemcpyHostToDevice, st));
      }
      if (run_prog) {
         HIPCHK(e->d_ops_tmp.ensure(prog.ops.size()));
         const Op *ho = e->stage.put(prog.ops.data(), prog.ops.size());
         HIPCHK(hipMemcpyAsync(e->d_ops_tmp.p, ho, prog.ops.size() * sizeof(Op), hipMemcpyHostToDevice, st));
      }
      HIPCHK(e->stage.end(st));
      if (run_pmat) {
         HIPCHK(e->d_rowmajor.ensure((size_t)psets * nn * n * n));
         HIPCHK(e->d_pint.ensure((size_t)psets * nn * 4096));
         HIPCHK(e->d_ptip.ensure((size_t)psets * nn * tip_words(e)));
         PmatArgs pa{};
         pa.n = n; pa.n_nodes = nn; pa.root = A; pa.K = K; pa.n_genes = G; pa.n_labels = e->n_labels;
         pa.n_codes = e->n_codes; pa.layout = 1;
         pa.label = e->d_label_eff.p; pa.is_leaf = e->d_is_leaf.p; pa.branch = e->d_branch.p; pa.rate = e->d_rate.p;
         pa.gene_rate = e->d_gene_rate.p; pa.eigen_of = e->d_eigen_of.p; pa.qfactor = e->d_qfactor.p;
         pa.eigen = e->d_eigen.p; pa.n_chara = e->d_n_chara.p; pa.chara_map = e->d_chara_map.p; pa.plain_codes = e->plain_codes;
         pa.rowmajor = e->d_rowmajor.p; pa.pint = e->d_pint.p; pa.ptip = e->d_ptip.p; pa.tip_words = (long)tip_words(e);
         HIPCHK(e->d_pcol.ensure((size_t)psets * nn * 64));
         pa.pcol = e->d_pcol.p;
         pa.B = 1; pa.rate_gs = e->rate_per_gene ? K : 0;
         InlineVec iv;
         iv.n_branch = iv.n_rate = 0;
         launch_pmat(pa, iv, nn, psets, false, st, pmat_on_matrix_cores(e, pa));
         e->n_pmat += (long)psets * (nn - 2);
         e->prog_valid = false;      // d_branch / P buffers now hold re-oriented edge data: the next eval rebuilds
         e->pmat_valid = false;
      }
      // A refill — every branch length moved since the partials were formed (minB's round after ming2 has moved kappa / omega, the first
      // call of a search): the forest of dirty subtrees is most of the tree, and the program is the same every time it happens at this
      // branch.  From the second time on it runs on a per-tree kernel of its own (round 6: STOREs in the resident layout, as a
      // keep-partials evaluation — 128-pattern tiles, the operand ring — instead of the 64-pattern interpreter), compiled on the worker
      // thread while the interpreter serves, or at once when the caller asked for per-tree kernels.
      bool refill_done = false;
      if (run_prog && e->jit_enabled && !e->env.force_gather && e->n_tips <= 207 && (e->n_codes <= 64 || e->amb_ascending)) {
         int n_store = 0;
         for (const Op &o : prog.ops) n_store += o.code == OP_STORE;
         Program full = prog;
         finish_program(full);
         if (2 * n_store >= n_int && jit_supported(full, e->n_tips, e->n_codes, e->n_pi, 6, 128, true)) {
            const std::string key = "b" + std::to_string(n) + "c" + std::to_string(e->n_codes) + ":" + jit_program_key(full, e->n_tips);
            bool have = e->jit_recall(key);
            if (!have && e->bjit_failed_key != key) {
               paml_amd_engine::JitJob *job = e->bjit_job.get();
               if (job && job->state.load() >= 2) {
                  if (job->th.joinable()) job->th.join();
                  if (job->state.load() == 2 && job->key == key) {
                     JitKernel nk;
                     if (jit_load_code(job->code, &nk) == 0) { nk.key = key; e->jit_retire(); e->jit = nk; have = true; }
                     else { if (nk.mod) (void)hipModuleUnload(nk.mod); e->bjit_failed_key = key; }
                  }
                  else if (job->key == key) { e->bjit_failed_key = key; e->err = "jit (refill): " + job->log; }
                  e->bjit_job.reset();
                  job = nullptr;
               }
               if (!have && !job && e->bjit_failed_key != key && (e->jit_forced || e->env.jit_sync || e->jit_count_request(key) >= 2)) {
                  const std::string src = jit_generate(full, e->n_tips, n, e->n_codes, 8);
                  std::vector<char> code;
                  if (jit_cached_code(src, &code) || ((e->jit_forced || e->env.jit_sync) && jit_compile_code(src, &code, &e->err) == 0)) {
                     JitKernel nk;
                     if (jit_load_code(code, &nk) == 0) { nk.key = key; e->jit_retire(); e->jit = nk; have = true; }
                     else { if (nk.mod) (void)hipModuleUnload(nk.mod); e->bjit_failed_key = key; }
                  }
                  else if (e->jit_forced || e->env.jit_sync) e->bjit_failed_key = key;
                  else {
                     e->bjit_job.reset(new paml_amd_engine::JitJob());
                     job = e->bjit_job.get();
                     job->key = key; job->src = src;
                     job->state.store(1);
                     job->th = std::thread([job]() { job->state.store(jit_compile_code(job->src, &job->code, &job->log) == 0 ? 2 : 3); });
                  }
               }
            }
            if (have) {
               e->use_jit = true;      // (kernel_name: the last pruning kernel was a per-tree one)
               if (int rc = select_tiles(e, true, 8, true)) return rc;
               const int n_blocks = e->n_tiles * K;
               int overflow = 0;
               if (full.max_stack > MFMA_RS) {
                  overflow = full.max_stack - MFMA_RS;
                  HIPCHK(e->d_stack.ensure((size_t)n_blocks * overflow * 8 * 1024));
               }
               if (T.n_scale) HIPCHK(e->d_fscale.ensure((size_t)K * e->n_patt));
               PruneArgs pr{};
               pr.ops = nullptr; pr.z = e->d_z.p; pr.z_stride = e->n_patt; pr.tiles = e->d_tiles.p; pr.n_tiles = e->n_tiles;
               pr.gene_off = e->d_gene_off.p; pr.weights = e->d_weights.p; pr.ztiles = e->d_ztiles.p; pr.zt_bytes = e->zt_bytes;
               pr.n = n; pr.n_tips = e->n_tips; pr.n_nodes = nn; pr.K = K; pr.n_genes = G; pr.n_codes = e->n_codes;
               pr.cleandata = e->cleandata; pr.n_pi = e->n_pi; pr.mode = e->mode; pr.n_scale = T.n_scale; pr.keep = 1; pr.n_patt = e->n_patt;
               pr.pi = e->d_pi.p; pr.pint = e->d_pint.p; pr.ptip = e->d_ptip.p; pr.pcol = e->d_pcol.p; pr.fscale = e->d_fscale.p;
               HIPCHK(e->d_fhK.ensure((size_t)K * e->n_patt));
               pr.fhK = e->d_fhK.p; pr.partials = e->d_bl_partials.p; pr.scalef = e->d_bl_scalef.p; pr.stack_scratch = e->d_stack.p;
               pr.stack_overflow_slots = overflow; pr.first_matmul = full.first_matmul; pr.n_int = n_int; pr.first_tip = full.first_tip;
               pr.tip_words = (long)tip_words(e); pr.tile_group0 = e->d_tile_group0.p; pr.part_groups = e->part_groups();
               pr.part_dump = e->d_bl_partials.p + words; pr.code_mask = e->d_code_mask.p;
               void *params[] = {&pr};
               HIPCHK(hipModuleLaunchKernel(e->jit.fn, std::min(n_blocks, e->n_cu), 1, 1, 8 * 64, 1, 1, 0, st, params, nullptr));
               refill_done = true;
               e->n_branch_refill_jit++;
            }
         }
      }
      if (run_prog && !refill_done) {
         const int n_blocks = e->n_tiles_full * K;
         int overflow = 0;
         if (prog.max_stack > MFMA_RS) {
            overflow = prog.max_stack - MFMA_RS;
            HIPCHK(e->d_stack.ensure((size_t)n_blocks * overflow * GATHER_WAVES * 1024));
         }
         PruneArgs pr{};
         pr.ops = e->d_ops_tmp.p; pr.z = e->d_z.p; pr.z_stride = e->n_patt; pr.tiles = e->d_tiles_full.p; pr.n_tiles = e->n_tiles_full;
         pr.gene_off = e->d_gene_off.p; pr.weights = e->d_weights.p;
         pr.n = n; pr.n_tips = e->n_tips; pr.n_nodes = nn; pr.K = K; pr.n_genes = G; pr.n_codes = e->n_codes;
         pr.cleandata = e->cleandata; pr.n_pi = e->n_pi; pr.mode = e->mode; pr.n_scale = T.n_scale; pr.keep = 1; pr.n_patt = e->n_patt;
         pr.pi = e->d_pi.p; pr.pint = e->d_pint.p; pr.ptip = e->d_ptip.p;
         HIPCHK(e->d_fhK.ensure((size_t)K * e->n_patt));
         pr.fhK = e->d_fhK.p; pr.partials = e->d_bl_partials.p; pr.scalef = e->d_bl_scalef.p; pr.stack_scratch = e->d_stack.p;
         pr.stack_overflow_slots = overflow; pr.first_matmul = prog.first_matmul; pr.n_int = n_int;
         pr.first_tip = -1; pr.tip_words = (long)tip_words(e); pr.part_groups = e->part_groups();
         launch_prune_full(e, prog.max_stack, n_blocks, pr, st);
      }
      EigPrepArgs ea{};
      ea.n = n; ea.K = K; ea.n_labels = e->n_labels; ea.n_t = n_t; ea.label = T.label[node_b]; ea.n_codes = e->n_codes;
      ea.rate_gs = e->rate_per_gene ? K : 0; ea.only_etab = (hit || bc.frag_ok[lab_b]) ? 1 : 0;      // (the operand-order matrices depend on the eigen systems only)
      bc.frag_ok[lab_b] = 1;
      ea.t = e->d_tt.p; ea.rate = e->d_rate.p; ea.gene_rate = e->d_gene_rate.p; ea.qfactor = e->d_qfactor.p; ea.pi = e->d_pi_plain.p;
      ea.eigen_of = e->d_eigen_of.p; ea.eigen = e->d_eigen.p; ea.code_mask = e->d_code_mask.p;
      ea.efrag = efrag; ea.ztab = ztab; ea.etab = e->d_bl_etab.p; ea.ecol = n == 61 ? e->d_bl_ecol.p + (size_t)lab_b * K * 128 : nullptr;
      hipLaunchKernelGGL(branch_eigprep_kernel, dim3(K), dim3(256), 0, st, ea);
      // (timing experiments of profiles/r04_branch.txt — NOFEVAL, and NOSTORE / NOMFMA whose results are garbage — exist only in a library
      //  built with PAML_AMD_EXTRA_FLAGS=-DPAML_AMD_BEIG_EXPERIMENTS; the production library does not read these variables)
#ifdef PAML_AMD_BEIG_EXPERIMENTS
      static const bool exp_nofeval = getenv("PAML_AMD_BEIG_NOFEVAL") != nullptr;
#else
      const bool exp_nofeval = false;
#endif
      const bool feval = !hit && K == 1 && n_t <= BEIG_NT && !exp_nofeval;
      e->bk_timed = false;
      if (e->profiling) {
         for (hipEvent_t &ev : e->ev_bk)
            if (!ev) HIPCHK(hipEventCreate(&ev));
         HIPCHK(hipEventRecord(e->ev_bk[0], st));
      }
      if (!hit) {
         BranchEigArgs ba{};
         ba.n = n; ba.K = K; ba.n_patt = e->n_patt; ba.n_tips = e->n_tips; ba.n_int = n_int; ba.n_nodes = nn; ba.n_groups = n_groups;
         ba.n_scale = T.n_scale; ba.n_t = n_t; ba.n_codes = e->n_codes; ba.a_node = A; ba.b_node = Bn;
         ba.n_sons = n_sons; ba.son[0] = son[0]; ba.son[1] = son[1]; ba.feval = feval ? 1 : 0;
         ba.chunk_groups = cg; ba.nb_local = nb_local; ba.first_chunk = e->first_chunk; ba.n_out = n_out; ba.n_rows = nrows;
         ba.partials = e->d_bl_partials.p; ba.scalef = scaled ? e->d_bl_scalef.p : nullptr; ba.z = e->d_z.p;
         ba.pint = e->d_pint.p; ba.ptip = e->d_ptip.p; ba.tip_words = (long)tip_words(e);
         ba.efrag = efrag; ba.ztab = ztab; ba.etab = e->d_bl_etab.p;
         ba.ecol = e->d_bl_ecol.p + (size_t)lab_b * K * 128; ba.pcol = e->d_pcol.p;
#ifdef PAML_AMD_BEIG_EXPERIMENTS
         static const int exp_abl = (getenv("PAML_AMD_BEIG_NOSTORE") ? 1 : 0) | (getenv("PAML_AMD_BEIG_NOMFMA") ? 2 : 0);
         ba.no_store = exp_abl;
#endif
         ba.freqK = e->d_freqK.p; ba.weights = e->d_weights.p; ba.coef = e->d_bl_coef.p; ba.partial = e->d_bpartial.p;
         const bool i0 = n_sons > 0 && !T.is_leaf(son[0]), i1 = n_sons > 1 && !T.is_leaf(son[1]);
         const int variant = n_sons == 0 ? 0 : (n_sons == 1 ? (i0 ? 1 : 2) : (i1 ? 3 : (i0 ? 4 : 5)));      // (two sons: the internal one, if any, comes first)
         beig_fn const fn = beig_kernels[variant][b_tip ? 1 : 0][n == 61 ? 1 : 0];
         hipLaunchKernelGGL(fn, dim3(std::min(nb_local, e->n_cu), K), dim3(512), BEIG_LDS_BYTES, st, ba);
         for (int v = e->n_tips; v < nn; v++) { bc.up[v] = up[v]; bc.ok[v] = 1; }
         e->n_branch_nodes += (long)std::count(clean.begin() + e->n_tips, clean.end(), 0);
         bc.coef_ok = true;
         bc.coef_node = node_b;
      }
      else e->n_branch_coef_hits++;
      if (!feval) {
         BranchPolyArgs pa{};
         pa.K = K; pa.n_patt = e->n_patt; pa.n_groups = n_groups; pa.n_scale = T.n_scale; pa.n_t = n_t;
         pa.chunk_groups = cg; pa.nb_local = nb_local; pa.first_chunk = e->first_chunk; pa.n_out = n_out; pa.n_rows = nrows;
         pa.coef = e->d_bl_coef.p; pa.etab = e->d_bl_etab.p; pa.scalef = scaled ? e->d_bl_scalef.p : nullptr; pa.weights = e->d_weights.p;
         pa.partial = e->d_bpartial.p;
         for (int it0 = 0; it0 < n_t; it0 += BEIG_NT) {
            pa.it0 = it0; pa.nt_here = std::min(BEIG_NT, n_t - it0);
            const size_t lds = ((size_t)K * pa.nt_here * 192 + 8 * 3 * BEIG_NT) * 8;
            hipLaunchKernelGGL(branch_poly_kernel, dim3(std::min(nb_local, 4 * e->n_cu)), dim3(512), lds, st, pa);
         }
      }
      HIPCHK(hipGetLastError());
      if (e->profiling) { HIPCHK(hipEventRecord(e->ev_bk[1], st)); e->bk_timed = true; }
      if (e->comm) {
         HIPCHK(hipEventRecord(e->ev_part[0], st));
         HIPCHK(hipStreamWaitEvent(e->sc, e->ev_part[0], 0));
         const ncclResult_t nr = rccl().AllReduce(e->d_bpartial.p, e->d_bpartial.p, (size_t)nrows * n_out, ncclDouble, ncclSum, e->comm, e->sc);
         if (nr != ncclSuccess) return fail(e, PAML_AMD_EHIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(nr));
         HIPCHK(hipEventRecord(e->ev_done[0], e->sc));
         HIPCHK(hipStreamWaitEvent(st, e->ev_done[0], 0));
      }
      if (int r = ensure_hout(e, (size_t)n_out)) return r;
      hipLaunchKernelGGL(branch_total_kernel, dim3(n_out), dim3(256), 0, st, (const double *)e->d_bpartial.p, nrows, n_out, e->h_out);      // (pinned, device-visible: no copy)
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(st));      // the one host synchronisation of the call
      if (int rc = eigen_fail_check(e)) return rc;
      for (int i = 0; i < n_t; i++) { lnL[i] = e->h_out[3 * i]; dlnL[i] = e->h_out[3 * i + 1]; ddlnL[i] = e->h_out[3 * i + 2]; }
      e->n_branch_eval++;
      cache_guard.ok = true;
      return 0;
   }
   bc.coef_ok = false;      // (the P / dP / ddP form below recomputes partials without the coefficients)
   if (any_dirty) {
      // the dirty partials: one program per side, run back to back in one launch of the full-featured kernels
      Program prog;
      for (int side = 0; side < 2; side++) {
         const int rt = side ? Bn : A;
         if (T.is_leaf(rt) || clean[rt]) continue;
         tr.root = rt;
         Program ps = build_program(tr, true, clean.data());
         for (const Op &o : ps.ops)
            if (o.code != OP_ROOT && o.code != OP_END) prog.ops.push_back(o);
         prog.max_stack = std::max(prog.max_stack, ps.max_stack);
         if (prog.first_matmul < 0) prog.first_matmul = ps.first_matmul;
      }
      prog.ops.push_back({OP_END, 0, 0, -1});
      // (prefetch links of the concatenated program: every MATMUL names the next one)
      {
         int next = -1;
         for (int i = (int)prog.ops.size() - 1; i >= 0; i--)
            if (prog.ops[i].code == OP_MATMUL || prog.ops[i].code == OP_MATMUL_POP) { prog.ops[i].c = next; next = prog.ops[i].a; }
         prog.first_matmul = next;
      }
      const int maxd = e->kk == KK_VALU20 ? VALU_MAXD_20 : VALU_MAXD_SMALL;
      if (!mfma && prog.max_stack > maxd) return fail(e, PAML_AMD_EUNSUPPORTED, "tree needs a deeper partial stack than this kernel provides");
      // P(t) of every edge in its new orientation
      HIPCHK(upload(e->d_label_eff, lab_eff.data(), lab_eff.size(), st));
      HIPCHK(upload(e->d_branch, br_eff.data(), br_eff.size(), st));
      HIPCHK(e->d_rowmajor.ensure((size_t)psets * nn * n * n));
      if (mfma) HIPCHK(e->d_pint.ensure((size_t)psets * nn * 4096));
      HIPCHK(e->d_ptip.ensure((size_t)psets * nn * tip_words(e)));
      PmatArgs pa{};
      pa.n = n; pa.n_nodes = nn; pa.root = A; pa.K = K; pa.n_genes = G; pa.n_labels = e->n_labels;
      pa.n_codes = e->n_codes; pa.layout = mfma ? 1 : 0;
      pa.label = e->d_label_eff.p; pa.is_leaf = e->d_is_leaf.p; pa.branch = e->d_branch.p; pa.rate = e->d_rate.p;
      pa.gene_rate = e->d_gene_rate.p; pa.eigen_of = e->d_eigen_of.p; pa.qfactor = e->d_qfactor.p;
      pa.eigen = e->d_eigen.p; pa.n_chara = e->d_n_chara.p; pa.chara_map = e->d_chara_map.p; pa.plain_codes = e->plain_codes;
      pa.rowmajor = e->d_rowmajor.p; pa.pint = e->d_pint.p; pa.ptip = e->d_ptip.p; pa.tip_words = (long)tip_words(e);
      pa.B = 1; pa.rate_gs = e->rate_per_gene ? K : 0;
      {
         InlineVec iv;
         iv.n_branch = iv.n_rate = 0;
         launch_pmat(pa, iv, nn, psets, false, st, pmat_on_matrix_cores(e, pa));
      }
      e->n_pmat += (long)psets * (nn - 2);
      e->prog_valid = false;      // d_branch / P buffers now hold re-oriented edge data: the next eval rebuilds
      e->pmat_valid = false;
      HIPCHK(upload(e->d_ops_tmp, prog.ops.data(), prog.ops.size(), st));
      const int n_blocks = e->n_tiles_full * K;
      int overflow = 0;
      if (mfma && prog.max_stack > MFMA_RS) {
         overflow = prog.max_stack - MFMA_RS;
         HIPCHK(e->d_stack.ensure((size_t)n_blocks * overflow * GATHER_WAVES * 1024));
      }
      PruneArgs pr{};
      pr.ops = e->d_ops_tmp.p; pr.z = e->d_z.p; pr.z_stride = e->n_patt; pr.tiles = e->d_tiles_full.p; pr.n_tiles = e->n_tiles_full;
      pr.gene_off = e->d_gene_off.p; pr.weights = e->d_weights.p;
      pr.n = n; pr.n_tips = e->n_tips; pr.n_nodes = nn; pr.K = K; pr.n_genes = G; pr.n_codes = e->n_codes;
      pr.cleandata = e->cleandata; pr.n_pi = e->n_pi; pr.mode = e->mode; pr.n_scale = T.n_scale; pr.keep = 1; pr.n_patt = e->n_patt;
      pr.pi = e->d_pi.p; pr.pint = mfma ? e->d_pint.p : e->d_rowmajor.p; pr.ptip = e->d_ptip.p;
      pr.fhK = e->d_fhK.p; pr.partials = e->d_bl_partials.p; pr.scalef = e->d_bl_scalef.p; pr.stack_scratch = e->d_stack.p;
      pr.stack_overflow_slots = overflow; pr.first_matmul = prog.first_matmul; pr.n_int = n_int;
      pr.first_tip = -1; pr.tip_words = (long)tip_words(e); pr.part_groups = e->part_groups();
      launch_prune_full(e, prog.max_stack, n_blocks, pr, st);
      HIPCHK(hipGetLastError());
      for (int v = e->n_tips; v < nn; v++) { bc.up[v] = up[v]; bc.ok[v] = 1; }
      e->n_branch_nodes += (long)std::count(clean.begin() + e->n_tips, clean.end(), 0);
   }

   // P, dP, ddP for every trial length, then the per-pattern contraction and the three weighted sums
   std::vector<double> tt(t, t + n_t);
   HIPCHK(upload(e->d_tt, tt.data(), tt.size(), st));
   HIPCHK(e->d_deriv.ensure((size_t)psets * n_t * 3 * n * n));
   if (mfma) HIPCHK(e->d_bl_frag.ensure((size_t)psets * n_t * 3 * 4096));
   DerivArgs da{};
   da.n = n; da.K = K; da.n_genes = G; da.n_labels = e->n_labels; da.n_t = n_t; da.label = T.label[node_b];
   da.rate_gs = e->rate_per_gene ? K : 0;
   da.t = e->d_tt.p; da.rate = e->d_rate.p; da.gene_rate = e->d_gene_rate.p; da.qfactor = e->d_qfactor.p;
   da.eigen_of = e->d_eigen_of.p; da.eigen = e->d_eigen.p; da.out = e->d_deriv.p; da.frag = mfma ? e->d_bl_frag.p : nullptr;
   hipLaunchKernelGGL(pmat_deriv_kernel, dim3(n_t, psets), dim3(256), 0, st, da);
   HIPCHK(e->d_bout.ensure((size_t)n_t * 3));
   // The 3 n_t sums (lnL, dlnL, ddlnL per trial length) are formed like the evaluation's total: one partial per block of patterns
   // at the block's GLOBAL position, the ranks' (disjoint, zero elsewhere) arrays summed over RCCL, then one fixed-order pass —
   // the same bits whatever the number of ranks.  Blocks: 64 patterns (matrix-core contraction) or 256.
   const int blk = mfma ? 64 : 256, n_out = 3 * n_t;
   const long nb_local = mfma ? e->n_tiles_full : (e->n_patt + 255) / 256;
   const bool sharded = e->comm != nullptr || e->n_patt_global != e->n_patt;      // (also: shard geometry without a communicator, for tests)
   const long nbg = sharded ? (e->n_patt_global + blk - 1) / blk : nb_local, fb = sharded ? e->first_patt / blk : 0;
   HIPCHK(e->d_bpartial.ensure((size_t)nbg * n_out));
   e->bpart_rows = nbg; e->bpart_cols = n_out; e->bpart_colmajor = false;
   if (sharded) HIPCHK(hipMemsetAsync(e->d_bpartial.p, 0, (size_t)nbg * n_out * sizeof(double), st));
   double *const bpart = e->d_bpartial.p + (size_t)fb * n_out;
   if (mfma) {
      const int nb = e->n_tiles_full;
      BranchMfmaArgs ba{};
      ba.n = n; ba.K = K; ba.n_genes = G; ba.n_patt = e->n_patt; ba.n_pi = e->n_pi; ba.n_tips = e->n_tips; ba.n_int = n_int;
      ba.n_tiles = nb; ba.n_scale = T.n_scale; ba.n_t = n_t; ba.a_node = A; ba.b_node = Bn;
      ba.tiles = e->d_tiles_full.p; ba.gene_off = e->d_gene_off.p; ba.partials = e->d_bl_partials.p;
      ba.scalef = scaled ? e->d_bl_scalef.p : nullptr; ba.zb = b_tip ? e->d_z.p + (size_t)Bn * e->n_patt : nullptr;
      ba.code_mask = e->d_code_mask.p; ba.pi = e->d_pi.p; ba.freqK = e->d_freqK.p; ba.weights = e->d_weights.p;
      ba.frag = e->d_bl_frag.p; ba.partial = bpart;
      for (int it = 0; it < n_t; it++) {
         ba.it = it;
         hipLaunchKernelGGL(branch_mfma_kernel, dim3(nb), dim3(256), 0, st, ba);
      }
   }
   else {
      const int nb = (e->n_patt + 255) / 256;
      BranchArgs ba{};
      ba.n = n; ba.K = K; ba.n_genes = G; ba.n_patt = e->n_patt; ba.n_t = n_t; ba.n_pi = e->n_pi; ba.b_is_tip = b_tip ? 1 : 0;
      ba.n_codes = e->n_codes; ba.cls_stride = (long)n_int * e->n_patt * n;
      ba.A = e->d_bl_partials.p + (size_t)(A - e->n_tips) * e->n_patt * n;
      ba.B = b_tip ? nullptr : e->d_bl_partials.p + (size_t)(Bn - e->n_tips) * e->n_patt * n;
      ba.SA = scaled ? e->d_bl_scalef.p : nullptr; ba.SB = nullptr; ba.n_scale = T.n_scale;
      ba.zb = b_tip ? e->d_z.p + (size_t)Bn * e->n_patt : nullptr;
      ba.n_chara = e->d_n_chara.p; ba.chara_map = e->d_chara_map.p; ba.freqK = e->d_freqK.p;
      ba.weights = e->d_weights.p; ba.PdP = e->d_deriv.p; ba.gene_off = e->d_gene_off.p; ba.partial = bpart;
      ba.pi = e->d_pi_plain.p;
      hipLaunchKernelGGL(branch_kernel, dim3(nb), dim3(256), 0, st, ba);
   }
   HIPCHK(hipGetLastError());
   if (e->comm) {      // the exchange step of the branch-local evaluation, on the communicator's own stream like every collective
      HIPCHK(hipEventRecord(e->ev_part[0], st));
      HIPCHK(hipStreamWaitEvent(e->sc, e->ev_part[0], 0));
      const ncclResult_t nr = rccl().AllReduce(e->d_bpartial.p, e->d_bpartial.p, (size_t)nbg * n_out, ncclDouble, ncclSum, e->comm, e->sc);
      if (nr != ncclSuccess) return fail(e, PAML_AMD_EHIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(nr));
      HIPCHK(hipEventRecord(e->ev_done[0], e->sc));
      HIPCHK(hipStreamWaitEvent(st, e->ev_done[0], 0));
   }
   hipLaunchKernelGGL(branch_reduce_kernel, dim3(1), dim3(256), 0, st, (const double *)e->d_bpartial.p, (int)nbg, n_out, e->d_bout.p);
   HIPCHK(hipGetLastError());
   {
      int r = ensure_hout(e, (size_t)n_t * 3);
      if (r) return r;
   }
   HIPCHK(hipMemcpyAsync(e->h_out, e->d_bout.p, (size_t)n_t * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
   HIPCHK(hipStreamSynchronize(st));      // the one host synchronisation of the call
   if (int rc = eigen_fail_check(e)) return rc;
   for (int i = 0; i < n_t; i++) { lnL[i] = e->h_out[3 * i]; dlnL[i] = e->h_out[3 * i + 1]; ddlnL[i] = e->h_out[3 * i + 2]; }
   e->n_branch_eval++;
   cache_guard.ok = true;
   return 0;
}

int paml_amd_branch_counters(const paml_amd_engine *e, long *n_calls, long *n_nodes_recomputed)
{
   if (!e) return PAML_AMD_EINVAL;
   if (n_calls) *n_calls = e->n_branch_eval;
   if (n_nodes_recomputed) *n_nodes_recomputed = e->n_branch_nodes;
   return 0;
}

long paml_amd_branch_coef_hits(const paml_amd_engine *e) { return e ? e->n_branch_coef_hits : -1; }

long paml_amd_branch_refill_kernels(const paml_amd_engine *e) { return e ? e->n_branch_refill_jit : -1; }

double paml_amd_branch_kernel_ms(paml_amd_engine *e)
{
   float ms = -1;
   if (!e || !e->bk_timed || hipEventElapsedTime(&ms, e->ev_bk[0], e->ev_bk[1]) != hipSuccess) { (void)hipGetLastError(); return -1; }
   return ms;
}

int paml_amd_get_branch_partials(paml_amd_engine *e, double *out, long cap, long *rows, int *cols)
{
   if (!e || !rows || !cols) return PAML_AMD_EINVAL;
   *rows = e->bpart_rows; *cols = e->bpart_cols;
   if (!out) return 0;
   if (cap < e->bpart_rows * e->bpart_cols || !e->d_bpartial.p) return fail(e, PAML_AMD_EINVAL, "get_branch_partials: no branch evaluation yet, or the buffer is too small");
   if (!e->bpart_colmajor) { HIPCHK(hipMemcpy(out, e->d_bpartial.p, (size_t)e->bpart_rows * e->bpart_cols * sizeof(double), hipMemcpyDeviceToHost)); }
   else {
      std::vector<double> t((size_t)e->bpart_rows * e->bpart_cols);
      HIPCHK(hipMemcpy(t.data(), e->d_bpartial.p, t.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (long r = 0; r < e->bpart_rows; r++)
         for (int c = 0; c < e->bpart_cols; c++) out[r * e->bpart_cols + c] = t[(size_t)c * e->bpart_rows + r];
   }
   return 0;
}

int paml_amd_node_posterior(paml_amd_engine *e, int node, const double *branch, const double *gene_rate, double *post)
{
   enter(e);
   if (!e || !branch || !post) return fail(e, PAML_AMD_EINVAL, "node_posterior: null argument");
   if (!(e->have_tips && e->have_tree && e->have_pi && e->have_classes) || e->eigen.empty())
      return fail(e, PAML_AMD_EINVAL, "node_posterior before set_tips/set_tree/set_pi/set_classes/set_eigen");
   const TreeDesc &T = e->tree;
   const int nn = T.n_nodes, n = e->n, K = e->K;
   if (node < 0 || node >= nn || T.is_leaf(node)) return fail(e, PAML_AMD_EINVAL, "node_posterior: not an internal node");
   for (size_t i = 0; i < e->eigen.size(); i++)
      if (e->eigen[i].kind == PAML_AMD_EIGEN_QMAT)
         return fail(e, PAML_AMD_EUNSUPPORTED, "node_posterior: moving the root needs a reversible model");
   TreeDesc tr;
   int r = rerooted_pmat(e, node, -1, branch, gene_rate, &tr);
   if (r) return r;
   Program prog = build_program(tr, false, nullptr);
   for (Op &o : prog.ops)
      if (o.code == OP_ROOT) o.code = OP_EXPORT;
   const bool scaled = T.n_scale > 0;
   HIPCHK(e->d_expA.ensure((size_t)K * e->n_patt * n));
   if (scaled) HIPCHK(e->d_expSA.ensure((size_t)K * e->n_patt));
   r = run_prune_full(e, prog, e->d_expA.p, scaled ? e->d_expSA.p : nullptr);
   if (r) return r;
   HIPCHK(e->d_expB.ensure((size_t)e->n_patt * n));
   PostArgs pa{};
   pa.n = n; pa.K = K; pa.n_genes = e->n_genes; pa.n_patt = e->n_patt; pa.n_pi = e->n_pi;
   pa.L = e->d_expA.p; pa.S = scaled ? e->d_expSA.p : nullptr; pa.pi = e->d_pi_plain.p; pa.freqK = e->d_freqK.p;
   pa.gene_off = e->d_gene_off.p; pa.post = e->d_expB.p;
   hipLaunchKernelGGL(posterior_kernel, dim3((e->n_patt + 255) / 256), dim3(256), 0, e->stream, pa);
   HIPCHK(hipGetLastError());
   HIPCHK(hipMemcpyAsync(post, e->d_expB.p, (size_t)e->n_patt * n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return eigen_fail_check(e);
}

}  // extern "C"
