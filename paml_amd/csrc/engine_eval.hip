// engine_eval.hip — one likelihood evaluation: batched P(t) (Kernel A), the fused pruning kernel chosen for the problem (Kernel B:
// per-tree specialised, streamed or full interpreter), the reduction (Kernel C) and the exchange step over the ranks; and the
// entry points built on it (paml_amd_eval, _eval_batch, _eval_device, _eval_dirty, _eval_adg).
// Built for gfx950 only (one of the translation units of libpaml_amd.so, see engine_state.h).
#include "engine_state.h"
#include "kernels_pmat.h"
#include "kernels_prune.h"
#include "kernels_reduce.h"

namespace paml_amd {

int build_tiles(paml_amd_engine *e)
{
   std::vector<int2> tiles;
   for (int g = 0; g < e->n_genes; g++)
      for (int h = e->gene_off[g]; h < e->gene_off[g + 1]; h += e->tile_patt) tiles.push_back(make_int2(g, h));
   e->n_tiles = (int)tiles.size();
   HIPCHK(upload(e->d_tiles, tiles.data(), tiles.size(), e->stream));
   const int tf = e->kk == KK_MFMA64 ? GATHER_WAVES * 16 : 256;
   std::vector<int2> tfull;
   for (int g = 0; g < e->n_genes; g++)
      for (int h = e->gene_off[g]; h < e->gene_off[g + 1]; h += tf) tfull.push_back(make_int2(g, h));
   e->n_tiles_full = (int)tfull.size();
   HIPCHK(upload(e->d_tiles_full, tfull.data(), tfull.size(), e->stream));
   if (e->kk == KK_MFMA64) {      // where a tile's resident partials live: the groups of the 64-pattern tiles it covers (consecutive within a gene)
      std::vector<int> g0(tiles.size());
      size_t k = 0;
      int base = 0;                // 64-pattern tiles of the genes before
      for (int g = 0; g < e->n_genes; g++) {
         for (int h = e->gene_off[g]; h < e->gene_off[g + 1]; h += e->tile_patt) g0[k++] = (base + (h - e->gene_off[g]) / tf) * GATHER_WAVES;
         base += (e->gene_off[g + 1] - e->gene_off[g] + tf - 1) / tf;
      }
      HIPCHK(upload(e->d_tile_group0, g0.data(), g0.size(), e->stream));
   }
   if (e->kk == KK_MFMA64 && e->tile_patt >= 128 && e->d_z.p && e->d_weights.p) {   // code blocks of the specialised kernel
      // (trees of more than 207 tips: two halves per tile, the rows in the order the tree's walk consumes them — jit_zplan; the tree's
      //  program is known by the time a kernel with 128-pattern tiles has been chosen, and launch_eval comes back here when it changes)
      JitZPlan zp;
      const bool half = !e->prog.ops.empty() && (zp = jit_zplan(e->prog, e->n_tips, e->tile_patt)).half;
      e->zt_bytes = half ? zp.pieces * zp.ZP * 2048 : jit_zpieces(e->n_tips, e->tile_patt) * 2048;
      e->zt_key = half ? jit_program_key(e->prog, e->n_tips) : std::string();
      if (half) {      // (row -> tip, then row -> position in the block)
         std::vector<int> both(zp.tip_of);
         both.insert(both.end(), zp.row_at.begin(), zp.row_at.end());
         HIPCHK(upload(e->d_ztip_of, both.data(), both.size(), e->stream));
      }
      HIPCHK(e->d_ztiles.ensure((size_t)e->n_tiles * e->zt_bytes));
      hipLaunchKernelGGL(ztile_kernel, dim3(e->n_tiles), dim3(e->tile_patt), 0, e->stream, e->d_tiles.p, e->d_gene_off.p, e->d_z.p, (long)e->n_patt,
                         e->d_weights.p, e->n_tips, e->zt_bytes, e->d_ztiles.p, half ? (const int *)e->d_ztip_of.p : (const int *)nullptr,
                         half ? (const int *)e->d_ztip_of.p + (e->n_tips + 1) : (const int *)nullptr);
   }
   HIPCHK(hipStreamSynchronize(e->stream));
   e->tiles_built_for = e->tile_patt;
   return 0;
}

// The tile tables of the kernel about to run (21 .. 64 states): 128-pattern tiles with their code blocks (per-tree kernel, streaming
// interpreter) or 64-pattern ones.  A change of tile size swaps the current set with the stashed one (built once each, engine_state.h
// TileStash) — the resident partials have ONE layout whatever the tile size (PruneArgs::part_groups), they stay valid.
int select_tiles(paml_amd_engine *e, bool big_tiles, int want_waves, bool jit_ok)
{
   if (big_tiles != e->mfma_dma || want_waves != e->mfma_waves) {
      e->swap_tile_stash();
      e->mfma_dma = big_tiles;
      e->mfma_waves = want_waves;
      e->tile_patt = e->mfma_waves * 16;
      bool have = e->tiles_built_for == e->tile_patt && e->d_tiles.p;
      if (have && jit_ok && e->tile_patt >= 128 && (e->n_tips > 200 || !e->zt_key.empty())) {      // piece mode: rows in the program's order
         const JitZPlan zp = jit_zplan(e->prog, e->n_tips, e->tile_patt);
         have = (zp.half ? jit_program_key(e->prog, e->n_tips) : std::string()) == e->zt_key;
      }
      if (!have)
         if (int r = build_tiles(e)) return r;
   }
   else if (jit_ok && e->tile_patt >= 128 && (e->n_tips > 200 || !e->zt_key.empty())) {      // piece mode: the code blocks' rows follow the tree's program
      const JitZPlan zp = jit_zplan(e->prog, e->n_tips, e->tile_patt);
      if ((zp.half ? jit_program_key(e->prog, e->n_tips) : std::string()) != e->zt_key)
         if (int r = build_tiles(e)) return r;
   }
   return 0;
}

// The one-pattern-per-lane interpreter: the register-stack instantiation that fits the program, else the scratch one.
static void launch_valu(paml_amd_engine *e, int max_stack, int n_blocks, const PruneArgs &pr, hipStream_t s)
{
   const dim3 g(n_blocks), b(256);
   switch (e->kk) {
   case KK_VALU4:
      if (max_stack <= 4) hipLaunchKernelGGL((prune_valu<4, 4, true>), g, b, 0, s, pr);
      else hipLaunchKernelGGL((prune_valu<4, VALU_MAXD_SMALL>), g, b, 0, s, pr);
      break;
   case KK_VALU5:
      if (max_stack <= 4) hipLaunchKernelGGL((prune_valu<5, 4, true>), g, b, 0, s, pr);
      else hipLaunchKernelGGL((prune_valu<5, VALU_MAXD_SMALL>), g, b, 0, s, pr);
      break;
   default:      // 20 states: a register stack costs > 256 VGPRs (one wave per SIMD) and measures 2x slower than scratch
      hipLaunchKernelGGL((prune_valu<20, VALU_MAXD_20>), g, b, 0, s, pr);
      break;
   }
}

// 21..64 states in the mfma64 layout, every eigen system a (U, V, Root) one: four workgroups per matrix on the matrix cores
// (pmat_mfma_kernel; PAML_AMD_PMAT_MFMA=0: pmat_kernel_t<64>, one workgroup per matrix on the vector units)
bool pmat_on_matrix_cores(const paml_amd_engine *e, const PmatArgs &pa)
{
   static const bool off = getenv("PAML_AMD_PMAT_MFMA") && atoi(getenv("PAML_AMD_PMAT_MFMA")) == 0;
   bool ok = e->kk == KK_MFMA64 && pa.layout == 1 && e->n_codes <= 256 && !off;
   for (const EigenHost &h : e->eigen) ok = ok && (h.kind == PAML_AMD_EIGEN_UVROOT || h.kind < 0);      // (< 0: an id never set, referred to by nothing)
   return ok;
}

void launch_pmat(const PmatArgs &pa, const InlineVec &iv, int n_nodes, int psets, bool small, hipStream_t s, bool mfma)
{
   const int gx = (n_nodes + std::max(pa.npb, 1) - 1) / std::max(pa.npb, 1);
   if (mfma) {
      static const int nt_env = getenv("PAML_AMD_PMAT_NT") ? atoi(getenv("PAML_AMD_PMAT_NT")) : -1;      // (experiments: 0 / 1 for every launch)
      if (nt_env >= 0 ? nt_env != 0 : (long)n_nodes * psets >= 128) hipLaunchKernelGGL(pmat_mfma_kernel<true>, dim3(n_nodes, psets, 4), dim3(256), 0, s, pa, iv);
      else hipLaunchKernelGGL(pmat_mfma_kernel<false>, dim3(n_nodes, psets, 4), dim3(256), 0, s, pa, iv);
   }
   else if (small) hipLaunchKernelGGL(pmat_small_kernel, dim3((n_nodes * psets + 7) / 8), dim3(256), 0, s, pa, iv);
   else if (pa.n <= 32 && pa.layout != 1 && pa.layout != 3) hipLaunchKernelGGL(pmat_kernel_t<32>, dim3(gx, psets), dim3(256), 2 * 32 * 32 * sizeof(double), s, pa, iv);
   else hipLaunchKernelGGL(pmat_kernel_t<64>, dim3(gx, psets), dim3(256), 2 * 4096 * sizeof(double), s, pa, iv);
}

void launch_prune_full(paml_amd_engine *e, int max_stack, int n_blocks, const PruneArgs &pr, hipStream_t s)
{
   if (e->kk == KK_MFMA64) hipLaunchKernelGGL(prune_mfma64_gather<GATHER_WAVES>, dim3(n_blocks), dim3(GATHER_WAVES * 64), 0, s, pr);
   else launch_valu(e, max_stack, n_blocks, pr, s);
}

void launch_zpm(const unsigned char *z, long z_stride, int n_tips, int n_patt, int zw, unsigned int *out, hipStream_t s)
{
   hipLaunchKernelGGL(zpm_kernel, dim3((n_patt + 255) / 256), dim3(256), 0, s, z, z_stride, n_tips, n_patt, zw, out);
}

// The specialised kernel for `key`: reuse the loaded module or generate + compile + load it.  A compile failure is
// not fatal (the interpreter kernels take over) unless PAML_AMD_JIT_STRICT is set.
template <class GEN>
static int ensure_jit(paml_amd_engine *e, const std::string &key, GEN gen, bool *ok)
{
   *ok = false;
   if (e->jit_recall(key)) { *ok = true; return 0; }      // (the kernel in use, or one of this engine's other programs kept loaded)
   e->jit_retire();
   std::string log;
   const std::string src = gen();
   if (!e->env.jit_dump.empty()) {
      FILE *f = fopen(e->env.jit_dump.c_str(), "w");
      if (f) { fputs(src.c_str(), f); fclose(f); }
   }
   if (jit_compile(src, &e->jit, &log) == 0) {
      e->jit.key = key;
      *ok = true;
   }
   else {
      e->err = "jit: " + log;
      if (e->env.jit_strict) return fail(e, PAML_AMD_EHIP, e->err);
   }
   return 0;
}

int launch_eval(paml_amd_engine *e, const double *branch, const double *gene_rate, const unsigned char *clean,
                double *d_lnL_out, bool want_lnf, const BatchSpec *bs, bool want_pipe, bool want_fhk)
{
   if (!(e->have_tips && e->have_tree && e->have_pi && e->have_classes))
      return fail(e, PAML_AMD_EINVAL, "eval before set_tips/set_tree/set_pi/set_classes");
   if (e->eigen.empty()) return fail(e, PAML_AMD_EINVAL, "eval before any set_eigen_*");
   const bool keep = (e->flags & PAML_AMD_KEEP_PARTIALS) != 0;
   if (clean && (!keep || !e->partials_valid))
      return fail(e, PAML_AMD_EINVAL, "eval_dirty needs PAML_AMD_KEEP_PARTIALS and a previous full evaluation");
   const int B = bs ? bs->B : 1, Km = e->K;          // Km: classes of the model; K: classes the kernels see
   const int n = e->n, nn = e->tree.n_nodes, K = Km * B, G = e->n_genes;
   const int psets = G * K;
   if (B > 1 && (keep || clean)) return fail(e, PAML_AMD_EUNSUPPORTED, "eval_batch: not with PAML_AMD_KEEP_PARTIALS");

   // program (tree walk) — rebuilt when the tree or the clean set changes
   const bool new_prog = !e->prog_valid || clean;
   if (new_prog) {
      e->prog = build_program(e->tree, keep, clean);
      e->prog_valid = (clean == nullptr);
      const int maxd = e->kk == KK_VALU20 ? VALU_MAXD_20 : VALU_MAXD_SMALL;
      if (e->kk != KK_MFMA64 && e->prog.max_stack > maxd) {
         e->prog_valid = false;
         return fail(e, PAML_AMD_EUNSUPPORTED, "tree needs a deeper partial stack than this kernel provides");
      }
   }
   // the fast path of consecutive eval_device calls (see pipe_ok): nothing but branch lengths / gene rates may have changed
   // (worth its event traffic only where the pruning kernel is long: the 21..64-state kernels and the 20-state matrix-core kernel on
   //  >= 10^5 pattern-classes)
   want_pipe = want_pipe && (e->kk == KK_MFMA64 || (e->kk == KK_VALU20 && e->want_m20)) && (long)e->n_patt * e->K >= 100000;
   const bool pipe = want_pipe && e->pipe_ok && !bs && !clean && !keep && !new_prog && !e->eigen_dirty && !e->env.no_pipeline;
   // Two pruning streams (paml_amd_engine::sb): from the second evaluation of such a run on, the evaluations alternate between the
   // engine's stream and `sb`, so that the persistent workgroups of evaluation i + 1 take the CUs as those of evaluation i leave
   // them — no kernel boundary, reduction or half-empty last round of tiles between two pruning kernels.  lane = reduction slot.
   const bool dual = pipe && e->dual_ok && !e->profiling;
   const int lane = dual ? e->red_slot : 0;
   if (dual && lane && !e->sb[lane - 1]) HIPCHK(create_engine_stream(&e->sb[lane - 1]));
   if (dual)
      if (int rc = ensure_side_stream(e)) return rc;
   hipStream_t const ms = lane ? e->sb[lane - 1] : e->stream;      // the stream of this evaluation's pruning kernel and partial sums
   if (want_pipe && !e->s2) {
      for (hipEvent_t &ev : e->ev_setread) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      for (int i = 0; i < paml_amd_engine::NPSET - 1; i++) e->spare[i].id = i + 1;
      HIPCHK(create_engine_stream(&e->s2));
      HIPCHK(hipEventCreateWithFlags(&e->ev_entry[0], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e->ev_entry[1], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e->ev_pmat, hipEventDisableTiming));
   }
   hipStream_t ps = e->stream;                    // stream of the uploads and of the P(t) kernel
   if (pipe) {
      // the side stream may overwrite the other P set once everything the main stream held in front of the PREVIOUS pruning
      // kernel is done: that set's last reader (the kernel before it), and the previous evaluation's own uploads and P(t)
      ps = e->s2;
      // (two pruning streams: only the run's first such evaluation — the uploads in front of the run; after that the side stream
      //  waits for nothing but the last reader of the set it is about to overwrite, four evaluations back)
      if (e->have_prev_entry && !(dual && e->dual_run)) HIPCHK(hipStreamWaitEvent(e->s2, e->ev_entry[e->entry_sel ^ 1], 0));
      const int nid = e->spare[e->spare_head].id;
      if (dual && e->setread_rec[nid]) HIPCHK(hipStreamWaitEvent(e->s2, e->ev_setread[nid], 0));
   }
   const bool skip_entry = dual && e->dual_run;      // (nothing waits for this evaluation's entry event)
   e->dual_run = dual;
   std::vector<EigenDev> tab;
   if (e->eigen_dirty)
      if (int rc = eigen_table(e, tab)) return rc;
   if (!(bs && bs->eigen_of))
      if (int rc = eigen_refs_ok(e, e->h_eigen_of.data(), e->h_eigen_of.size(), "eval")) return rc;

   // branch lengths and gene rates of a single evaluation ride in the kernel arguments of P(t) (InlineVec): no copy at all
   InlineVec iv;
   iv.n_branch = iv.n_rate = 0;
   const bool use_inline = B == 1 && nn + G <= PMAT_INLINE_MAX;
   if (use_inline) {
      iv.n_branch = nn; iv.n_rate = G;
      memcpy(iv.v, branch, (size_t)nn * sizeof(double));
      for (int g = 0; g < G; g++) iv.v[nn + g] = gene_rate ? gene_rate[g] : 1.0;
   }
   // the other small inputs (and the batched ones) go through the pinned arena: async H2D, no host stall
   if (!use_inline || bs || !tab.empty() || new_prog) {
      const size_t L = (size_t)e->n_labels;
      const size_t need = (size_t)B * nn * 8 + (size_t)B * G * 8 + tab.size() * sizeof(EigenDev) +
                          (bs ? (size_t)B * (G * Km * L * 4 + Km * L * 8 + 2 * Km * 8) + 64 : 0) +
                          (new_prog ? e->prog.ops.size() * sizeof(Op) + e->prog.stream.size() * sizeof(int) : 0) + 256;
      HIPCHK(e->stage.begin(need));
      DevBuf<double> &dbr = pipe ? e->d2_branch : e->d_branch, &dgr = pipe ? e->d2_gene_rate : e->d_gene_rate;   // (the side stream has its own)
      e->bl_gr_sent = false;
      HIPCHK(dbr.ensure((size_t)B * nn));
      HIPCHK(dgr.ensure((size_t)B * G));
      if (!use_inline) {
         const double *hb = e->stage.put(branch, (size_t)B * nn);
         HIPCHK(hipMemcpyAsync(dbr.p, hb, (size_t)B * nn * 8, hipMemcpyHostToDevice, ps));
         std::vector<double> gr((size_t)B * G, 1.0);
         if (gene_rate) gr.assign(gene_rate, gene_rate + (size_t)B * G);
         const double *hg = e->stage.put(gr.data(), gr.size());
         HIPCHK(hipMemcpyAsync(dgr.p, hg, gr.size() * 8, hipMemcpyHostToDevice, ps));
      }
      if (bs) {      // per-element class tables
         if (bs->eigen_of) {
            const size_t cnt = (size_t)B * G * Km * L;
            if (int rc = eigen_refs_ok(e, bs->eigen_of, cnt, "eval_batch")) return rc;
            HIPCHK(e->d_b_eigen_of.ensure(cnt));
            const int *h = e->stage.put(bs->eigen_of, cnt);
            HIPCHK(hipMemcpyAsync(e->d_b_eigen_of.p, h, cnt * 4, hipMemcpyHostToDevice, e->stream));
         }
         const double *src[3] = {bs->qfactor, bs->freqK, bs->rate};
         DevBuf<double> *dst[3] = {&e->d_b_qfactor, &e->d_b_freqK, &e->d_b_rate};
         const size_t cnt[3] = {(size_t)B * Km * L, (size_t)B * Km, (size_t)B * Km * (e->rate_per_gene ? G : 1)};
         for (int i = 0; i < 3; i++)
            if (src[i]) {
               HIPCHK(dst[i]->ensure(cnt[i]));
               const double *h = e->stage.put(src[i], cnt[i]);
               HIPCHK(hipMemcpyAsync(dst[i]->p, h, cnt[i] * 8, hipMemcpyHostToDevice, e->stream));
            }
      }
      if (!tab.empty()) {
         HIPCHK(e->d_eigen.ensure(tab.size()));
         const EigenDev *ht = e->stage.put(tab.data(), tab.size());
         HIPCHK(hipMemcpyAsync(e->d_eigen.p, ht, tab.size() * sizeof(EigenDev), hipMemcpyHostToDevice, e->stream));
         e->eigen_dirty = false;
      }
      if (new_prog) {
         HIPCHK(e->d_ops.ensure(e->prog.ops.size()));
         const Op *ho = e->stage.put(e->prog.ops.data(), e->prog.ops.size());
         HIPCHK(hipMemcpyAsync(e->d_ops.p, ho, e->prog.ops.size() * sizeof(Op), hipMemcpyHostToDevice, e->stream));
         HIPCHK(e->d_stream.ensure(e->prog.stream.size() + 2));
         if (!e->prog.stream.empty()) {
            const int *hs = e->stage.put(e->prog.stream.data(), e->prog.stream.size());
            HIPCHK(hipMemcpyAsync(e->d_stream.p, hs, e->prog.stream.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
         }
      }
      HIPCHK(e->stage.end(ps));
   }

   // P(t) storage (pipelined: the set the previous evaluation did not use)
   if (pipe) {
      paml_amd_engine::PSet &sp = e->spare[e->spare_head];      // the set used longest ago
      std::swap(e->d_rowmajor, sp.rowmajor); std::swap(e->d_pint, sp.pint); std::swap(e->d_ptip, sp.ptip); std::swap(e->d_pcol, sp.pcol);
      std::swap(e->pset, sp.id);
      e->spare_head = (e->spare_head + 1) % (paml_amd_engine::NPSET - 1);
   }
   HIPCHK(e->d_rowmajor.ensure((size_t)psets * nn * n * n));
   if (e->kk == KK_MFMA64) HIPCHK(e->d_pint.ensure((size_t)psets * nn * 4096));
   if (e->kk == KK_VALU20 && e->want_m20) HIPCHK(e->d_pint.ensure((size_t)psets * nn * 400));      // (20 states on the matrix cores: the operand-order copies, pmat layout 2)
   if (e->kk == KK_MFMA64) HIPCHK(e->d_pcol.ensure((size_t)psets * nn * 64));
   HIPCHK(e->d_ptip.ensure((size_t)psets * nn * tip_words(e)));
   HIPCHK(e->d_fhK.ensure((size_t)K * e->n_patt));
   // kernel choice for the 21..64-state path:
   //   jit    — straight-line kernel specialised for this tree (jit.h), 128 patterns per workgroup
   //   stream — the interpreter over the same operand stream (lean programs only), 128 patterns per workgroup
   //   gather — the full interpreter (keep-partials STORE/LOAD, deep stacks, > MFMA_ZT tips, > 64 codes), 64 per workgroup
   if (e->kk == KK_MFMA64) {
      bool lean = e->prog.max_stack <= MFMA_RS && e->n_tips <= MFMA_ZT && e->n_codes <= 64 && !e->env.force_gather;
      // small data sets (at most a quarter of the CUs get a 128-pattern tile): the 64-pattern workgroups of the gather kernel —
      // one wave per SIMD, twice as many workgroups — finish a tile in 0.63 of the time (13 taxa x 79 codon patterns: 42 against 66 us,
      // a batched gradient of 25 evaluations 0.125 against 0.151 ms; profiles/r02_small_latency.jsonl)
      if ((e->n_patt + 127) / 128 <= e->n_cu / 4 && !e->env.force_stream) lean = false;
      for (const Op &o : e->prog.ops)
         if (o.code == OP_PUSH || o.code == OP_SCALE || o.code == OP_STORE || o.code == OP_LOAD) lean = false;
      bool jit_ok = false;
      // waves per workgroup of the per-tree kernel: 8 (two per SIMD, 128 patterns per tile).  PAML_AMD_JIT_WAVES=12 builds the
      // three-per-SIMD variant (192-pattern tiles, <= 168 VGPRs): measured SLOWER on MI355X (1.659 against 1.622 ms at C4, 4.73 against
      // 4.63 ms with three classes — 40 spilled dwords and a third more LDS / DMA traffic per step), kept as a generator parameter
      int jw = 8;
      if (e->env.jit_waves == 12 && jit_supported(e->prog, e->n_tips, e->n_codes, e->n_pi, 6, 192) && jit_zbuffers(e->n_tips, 192) == 2) jw = 12;
      // (more than 64 codes: the per-tree kernel sums the rows of a code's states in ascending order — e->amb_ascending, set_tips)
      // LOAD programs (paml_amd_eval_dirty: one per set of clean nodes) get a kernel too (round 6) — compiled on the worker thread from the
      // SECOND time a set is asked for (minbranches' walk repeats its sets cycle after cycle; a set seen once is not worth 0.5 s of compiler),
      // the interpreter serving meanwhile; code-block pieces follow one program's order, so trees beyond 207 tips keep the interpreter there
      bool has_load = false;
      for (const Op &o : e->prog.ops) has_load = has_load || o.code == OP_LOAD;
      if (e->jit_enabled && !e->env.force_gather && !(has_load && e->n_tips > 207) && (e->n_codes <= 64 || e->amb_ascending) && jit_supported(e->prog, e->n_tips, e->n_codes, e->n_pi, 6, jw * 16, true)) {
         const std::string key = "m" + std::to_string(n) + "c" + std::to_string(e->n_codes) + "w" + std::to_string(jw) + (jit_rowtail(n) ? "r:" : ":") + jit_program_key(e->prog, e->n_tips);
         // Large trees (> 120 ops: roughly more than 35 taxa): tens of thousands of instructions, many seconds of compiler time.  Unless the
         // caller asked to wait (PAML_AMD_JIT flag / PAML_AMD_JIT_SYNC), the kernel is built on a worker thread while the interpreter kernels
         // serve, and the engine changes over when the code object is there; one found on disk is loaded at once.  Round 5: the generator
         // cuts such a walk into basic blocks (jit_split_mode), which is what the hardware wants and makes the full build as quick as the
         // build without the three passes that are quadratic on one giant block (JIT_BIG_FLAGS) — that two-stage build (quick kernel at 0.63
         // of the FP64 peak first, the full one after) remains for PAML_AMD_JIT_SPLIT=0 / asm.
         const bool big = e->prog.ops.size() > 120;
         const bool background = (big || has_load) && !e->jit_forced && !e->env.jit_sync;
         const bool wanted = !has_load || e->jit_forced || e->env.jit_sync || e->jit_recall(key) || e->jit_count_request(key) >= 2;
         if (!wanted) jit_ok = false;
         else if (!background) {
            int r = ensure_jit(e, key, [&]() { return jit_strip_big(jit_generate(e->prog, e->n_tips, n, e->n_codes, jw)); }, &jit_ok);
            if (r) return r;
            if (jit_ok) { e->jit_stage = 2; e->jit.stage = 2; }
         }
         else {
            // (a code object is loaded beside the kernel in use and replaces it only when the load succeeded: a full build that does not
            //  load leaves the quick one serving, and its key is remembered so that it is not built again and again)
            auto load = [&](const std::vector<char> &code, int stage) {
               JitKernel nk;
               if (jit_load_code(code, &nk) != 0) {
                  if (nk.mod) (void)hipModuleUnload(nk.mod);
                  return false;
               }
               e->jit_retire();
               e->jit = nk;
               e->jit.key = key; e->jit.stage = stage; e->jit_stage = stage;
               return true;
            };
            (void)e->jit_recall(key);      // (one of this engine's other programs, kept loaded)
            e->jit_stage = (e->jit.fn && e->jit.key == key) ? e->jit.stage : 0;      // (another tree's kernel, or none: stage 0)
            paml_amd_engine::JitJob *job = e->jit_job.get();
            if (job && job->state.load() >= 2 && job->th.joinable()) job->th.join();
            if (job && job->state.load() >= 2) {
               if (job->key == key && job->state.load() == 2) {          // a code object is there: load it and change over
                  if (!load(job->code, job->stage)) {
                     (job->stage == 1 ? e->jit_failed_key : e->jit_stage2_failed_key) = key;
                     e->err = "jit: hipModuleLoadData failed";
                  }
               }
               else if (job->key == key) {                               // failed
                  (job->stage == 1 ? e->jit_failed_key : e->jit_stage2_failed_key) = key;
                  e->err = "jit: " + job->log;
               }
               e->jit_job.reset();
               job = nullptr;
            }
            auto may_build = [&](int stage) { return (stage == 1 ? e->jit_failed_key : e->jit_stage2_failed_key) != key; };
            // two-stage build (generators without block splits only): the quick kernel first, the full one replaces it; else the one build
            const bool two = jit_split_mode(e->prog.ops.size()) != 2 && !getenv("PAML_AMD_JIT_ONE_STAGE");
            const int next = (e->jit_stage == 0 && two && may_build(1)) ? 1 : 2;
            if (!job && e->jit_stage < 2 && may_build(next)) {
               const std::string quick = jit_generate(e->prog, e->n_tips, n, e->n_codes, jw), full = jit_strip_big(quick);
               std::vector<char> code;
               if (e->jit_stage == 0) {      // a code object on disk is loaded at once
                  if (may_build(2) && jit_cached_code(full, &code)) { if (!load(code, 2)) e->jit_stage2_failed_key = key; }
                  else if (next == 1 && quick != full && jit_cached_code(quick, &code)) { if (!load(code, 1)) e->jit_failed_key = key; }
               }
               const int st = (e->jit_stage == 0 && next == 1 && quick != full) ? 1 : 2;
               if (e->jit_stage < st && may_build(st)) {
                  e->jit_job.reset(new paml_amd_engine::JitJob());
                  job = e->jit_job.get();
                  job->key = key; job->stage = st; job->src = st == 1 ? quick : full;
                  job->state.store(1);
                  job->th = std::thread([job]() { job->state.store(jit_compile_code(job->src, &job->code, &job->log) == 0 ? 2 : 3); });
               }
            }
            jit_ok = e->jit.fn && e->jit.key == key;
         }
      }
      e->use_jit = jit_ok;
      const bool big_tiles = jit_ok || lean;
      const int want_waves = jit_ok ? jw : (lean ? DMA_WAVES : GATHER_WAVES);
      if (int r = select_tiles(e, big_tiles, want_waves, jit_ok)) return r;
   }
   if (e->kk != KK_MFMA64) {      // 4 / 5 / 20 states: the interpreter unrolled for this tree
      bool jit_ok = false, fused = false;
      // (20 states: the unrolled walk needs > 256 VGPRs and runs at one wave per SIMD, slower than the interpreter)
      if (e->jit_enabled && !keep && n <= 5 && jit_valu_supported(e->prog)) {
         // the fused form (classes inside, LDS tip tables, reduction in the epilogue) when the model fits it
         const ValuFusedPlan pl = jit_valu_fused_plan(e->prog, n, e->n_tips, e->n_codes, Km, e->chunk);
         // Several genes (round 6): the fused form exists (jit_generate_valu_fused with G > 1: one gene's tables at a time, refilled where a
         // workgroup's chunks cross into the next gene; same bits) and is NOT the default: on MI355X it measures no faster than the unfused
         // kernel + reduce_stage1, which serve genes for free — a workgroup is one (tile, class) there — 32 taxa x 10^5 patterns x Gamma-4 in 4
         // genes: one evaluation 0.050 ms fused against 0.030 unfused, a gradient's 122 evaluations in one launch 1.77 against 1.77 ms (one
         // gene, fused: 0.029 / 1.33).  Why the several-genes form of the same inner loop runs 1.6 x slower than the one-gene form OUTSIDE
         // the profiler (within 4 % of it under rocprofv3, equal instruction counts) was bisected to the table fill living inside the
         // chunk loop and not resolved: profiles/r06_genes_4state.txt.  PAML_AMD_VF_GENES=1 switches it on (tests, measurements).
         const bool vf_genes = getenv("PAML_AMD_VF_GENES") && atoi(getenv("PAML_AMD_VF_GENES")) != 0;      // (read per call: tests switch it)
         if (pl.ok && (G == 1 ? e->n_pi == 1 : (vf_genes && (e->n_pi == 1 || e->n_pi == G))) && e->d_zpm.p && !e->env.no_fused && !(G > 1 && n == 4 && e->env.mfma4) && G <= 64) {
            // 4 states: the matrix-core form (v_mfma_f64_4x4x4) is an experiment kept behind PAML_AMD_MFMA4=1 — same issue slots as
            // the FMA form (an FP64 MFMA of 256 MACs takes 16 cycles, sixteen v_fma_f64 of a wave 64) and four times the
            // integer work per pattern (a lane is a (state, pattern) pair): 0.32 of peak against 0.64, profiles/r02_valu_fused_shapes.txt
            const bool m4 = n == 4 && e->env.mfma4;
            int r = ensure_jit(e, std::string(m4 ? "m4" : "vf") + std::to_string(n) + "c" + std::to_string(e->n_codes) + "k" + std::to_string(Km) + "r" + std::to_string(pl.R) + "w" +
                                     std::to_string(pl.CW) + (pl.cherry ? "y" : "n") + (G > 1 ? "g" + std::to_string(G) + ":" : ":") + jit_program_key(e->prog, e->n_tips),
                               [&]() { return m4 ? jit_generate_mfma4(e->prog, e->n_tips, e->n_codes, Km, e->chunk)
                                                 : jit_generate_valu_fused(e->prog, n, e->n_tips, e->n_codes, Km, e->chunk, G); }, &jit_ok);
            e->fused_mfma4 = jit_ok && m4;
            if (r) return r;
            fused = jit_ok;
            e->fused_threads = 256 * pl.CW;
         }
         if (!jit_ok) {
            int r = ensure_jit(e, "v" + std::to_string(n) + ":" + jit_program_key(e->prog, e->n_tips),
                               [&]() { return jit_generate_valu(e->prog, n); }, &jit_ok);
            if (r) return r;
         }
      }
      e->m20 = false;
      if (e->want_m20 && !clean && jit_m20_supported(e->prog, e->n_tips, G)) {
         int r = ensure_jit(e, std::string(getenv("PAML_AMD_M20_W12") ? "m20w12c" : getenv("PAML_AMD_M20_HALF") ? "m20hc" : "m20c") + std::to_string(e->n_codes) + (G > 1 ? "g:" : ":") + jit_program_key(e->prog, e->n_tips), [&]() { return jit_generate_m20(e->prog, e->n_tips, e->n_codes, G > 1 ? 2 : 1); }, &jit_ok);
         if (r) return r;
         e->m20 = jit_ok;
      }
      e->use_jit = jit_ok;
      e->fused = fused;
   }
   const bool use_dma = e->mfma_dma;
   const int n_blocks = e->n_tiles * K;
   const int n_int = nn - e->n_tips;
   if (keep) {
      size_t words = e->kk == KK_MFMA64 ? (size_t)K * n_int * e->part_groups() * 1024 + (size_t)std::max(e->mfma_waves, 8) * 1024      // (+ PruneArgs::part_dump: a row per wave)
                                        : (size_t)K * n_int * e->n_patt * n;
      HIPCHK(e->d_partials.ensure(words));
      HIPCHK(e->d_scalef.ensure((size_t)K * std::max(1, e->tree.n_scale) * e->n_patt));
   }
   int overflow = 0;
   if (e->kk == KK_MFMA64 && e->prog.max_stack > MFMA_RS) {
      overflow = e->prog.max_stack - MFMA_RS;
      HIPCHK(e->d_stack.ensure((size_t)n_blocks * overflow * e->mfma_waves * 1024));
   }

   // Small data sets on the 21..64-state interpreters: every 16-pattern group can have a CU (ONE round: the kernels' LDS leaves room for
   // one workgroup per CU, and two rounds of a 25 us walk lose to one round of the gather kernel's 39) — four waves per group
   // (prune_mfma64_coop), and once the tree's own kernel is there, that one with the reduction inside (jit_generate_coop: one launch
   // after P(t) per evaluation).  Single engines only: with pattern shards the chunk sums go through the exchange step.
   bool coop = false, coopj = false;
   if (e->kk == KK_MFMA64 && !e->use_jit && !use_dma) {
      coop = !keep && !clean && !e->env.no_coop && e->tile_patt == 64 && e->prog.max_stack <= COOP_SLOTS &&
             (long)e->n_tiles * 4 * K <= (long)e->n_cu && !e->env.prof_ops.size();
      for (const Op &o : e->prog.ops) coop = coop && o.code != OP_STORE && o.code != OP_LOAD && o.code != OP_EXPORT;
      if (coop && e->coopj_enabled && !e->comm && e->nb_global == (e->n_patt + e->chunk - 1) / e->chunk && e->first_chunk == 0 &&
          jit_coop_supported(e->prog, e->n_tips, e->n_codes)) {
         const std::string key = "cj" + std::to_string(n <= 32 ? n : 64) + ":" + jit_program_key(e->prog, e->n_tips);
         if (e->jit_coop.fn && e->jit_coop.key == key) coopj = true;
         else if (e->coop_failed_key != key) {
            paml_amd_engine::JitJob *job = e->coop_job.get();
            if (job && job->state.load() >= 2 && job->th.joinable()) job->th.join();
            std::vector<char> code;
            bool have = false;
            if (job && job->state.load() >= 2) {      // a finished compilation: this tree's, or an earlier tree's
               if (job->state.load() == 2 && job->key == key) { code.swap(job->code); have = true; }
               else if (job->state.load() == 3 && job->key == key) { e->coop_failed_key = key; e->err = "jit (coop): " + job->log; }
               e->coop_job.reset();
               job = nullptr;
            }
            if (!have && !job && e->coop_failed_key != key) {
               const std::string src = jit_generate_coop(e->prog, e->n_tips, n);
               if (!e->env.jit_dump.empty())
                  if (FILE *f = fopen(e->env.jit_dump.c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
               if (jit_cached_code(src, &code)) have = true;
               else if (e->jit_forced || e->env.jit_sync) {
                  std::string log;
                  if (jit_compile_code(src, &code, &log) == 0) have = true;
                  else {
                     e->coop_failed_key = key; e->err = "jit (coop): " + log;
                     if (e->env.jit_strict) return fail(e, PAML_AMD_EHIP, e->err);
                  }
               }
               else {
                  e->coop_job.reset(new paml_amd_engine::JitJob());
                  job = e->coop_job.get();
                  job->key = key; job->src = src;
                  job->state.store(1);
                  job->th = std::thread([job]() { job->state.store(jit_compile_code(job->src, &job->code, &job->log) == 0 ? 2 : 3); });
               }
            }
            if (have) {
               if (e->jit_coop.mod) (void)hipModuleUnload(e->jit_coop.mod);
               e->jit_coop = JitKernel();
               if (jit_load_code(code, &e->jit_coop) == 0) { e->jit_coop.key = key; coopj = true; }
               else { e->coop_failed_key = key; e->err = "jit (coop): hipModuleLoadData failed"; }
            }
         }
      }
   }
   e->coop = coop; e->coopj = coopj;

   // Kernel A: batched P(t)
   PmatArgs pa{};
   pa.n = n; pa.n_nodes = nn; pa.root = e->tree.root; pa.K = Km; pa.n_genes = G; pa.n_labels = e->n_labels;
   pa.n_codes = e->n_codes; pa.layout = e->kk == KK_MFMA64 ? ((e->use_jit && jit_rowtail(n)) ? 3 : 1) : ((e->kk == KK_VALU20 && e->use_jit && e->m20) ? 2 : 0);
   pa.label = e->d_label.p; pa.is_leaf = e->d_is_leaf.p; pa.branch = pipe ? e->d2_branch.p : e->d_branch.p; pa.rate = e->d_rate.p;
   pa.gene_rate = pipe ? e->d2_gene_rate.p : e->d_gene_rate.p; pa.eigen_of = e->d_eigen_of.p; pa.qfactor = e->d_qfactor.p;
   pa.eigen = e->d_eigen.p; pa.n_chara = e->d_n_chara.p; pa.chara_map = e->d_chara_map.p; pa.plain_codes = e->plain_codes;
   pa.rowmajor = e->d_rowmajor.p; pa.pint = e->d_pint.p; pa.ptip = e->d_ptip.p; pa.tip_words = (long)tip_words(e);
   pa.B = B; pa.branch_bs = nn; pa.gene_rate_bs = G; pa.pcol = e->kk == KK_MFMA64 ? e->d_pcol.p : nullptr;
   if (bs && bs->eigen_of) { pa.eigen_of = e->d_b_eigen_of.p; pa.eigen_of_bs = (long)G * Km * e->n_labels; }
   if (bs && bs->qfactor) { pa.qfactor = e->d_b_qfactor.p; pa.qfactor_bs = (long)Km * e->n_labels; }
   pa.rate_gs = e->rate_per_gene ? Km : 0;
   // PAML_AMD_PMAT_NPB=n (experiment): in a run of evaluations, where P(t) is built ahead on the side stream with the few CUs the
   // pruning kernels leave it, n nodes per workgroup with the eigen vectors kept on chip from node to node.  Measured on MI355X with
   // n = 8: 20 states 0.1887 against 0.1884 ms per evaluation, 61 states SLOWER (0.211 against 0.203 ms at the 8-GPU shard size,
   // 1.547 against 1.534 at 10^6 patterns: fewer workgroups, each eight times longer, and the pruning kernel waits for the last)
   static const int npb_run = getenv("PAML_AMD_PMAT_NPB") ? atoi(getenv("PAML_AMD_PMAT_NPB")) : 1;
   pa.npb = pipe ? npb_run : 1;
   if (bs && bs->rate) { pa.rate = e->d_b_rate.p; pa.rate_bs = e->rate_per_gene ? (long)G * Km : Km; }
   const bool pmat_mfma = pmat_on_matrix_cores(e, pa);
   // ... on the matrix cores in the mfma64 layout nobody reads the row-major copy (paml_amd_get_pmat rebuilds P from the operand-order ones)
   static const bool want_rowmajor = getenv("PAML_AMD_PMAT_ROWMAJOR") != nullptr;
   if (pmat_mfma && pa.layout == 1 && !want_rowmajor) pa.rowmajor = nullptr;
   e->rowmajor_valid = pa.rowmajor != nullptr;
   e->pmat_B = B;
   // ... and single evaluations get label -> eigen_of -> eigen set -> U / V / Root resolved on the host (PmatArgs::res)
   if (pmat_mfma && !bs && use_inline && !e->rate_per_gene && !e->eigen.empty()) {
      if (!e->pres_valid) {
         std::vector<PmatRes> res((size_t)psets * nn);
         for (int g = 0; g < G; g++)
            for (int ir = 0; ir < Km; ir++)
               for (int v = 0; v < nn; v++) {
                  const int lab = e->tree.label.empty() ? 0 : e->tree.label[v];
                  const EigenHost &h = e->eigen[e->h_eigen_of[((size_t)g * Km + ir) * e->n_labels + lab]];
                  res[((size_t)g * Km + ir) * nn + v] = PmatRes{h.U.p, h.V.p, h.Root.p, e->class_rate[ir], e->h_qfactor[(size_t)ir * e->n_labels + lab],
                                                                e->tree.is_leaf(v) ? 1 : 0, 0};
               }
         HIPCHK(e->d_pres.ensure(res.size()));
         // (pageable source: staged by the runtime before the call returns; in stream order in front of the P(t) kernel)
         HIPCHK(hipMemcpyAsync(e->d_pres.p, res.data(), res.size() * sizeof(PmatRes), hipMemcpyHostToDevice, ps));
         e->pres_valid = true;
      }
      pa.res = e->d_pres.p;
   }
   mark_on(e, ps);
   bool small_pmat = e->kk != KK_MFMA64 && n <= 5;
   for (const EigenHost &h : e->eigen) small_pmat = small_pmat && h.kind != PAML_AMD_EIGEN_QMAT;      // (ids never set: kind < 0, fine)
   launch_pmat(pa, iv, nn, psets, small_pmat, ps, pmat_mfma);
   mark_on(e, ps);
   if (pipe) {      // the pruning kernel (main stream) starts when this P(t) is there
      HIPCHK(hipEventRecord(e->ev_pmat, e->s2));
      HIPCHK(hipStreamWaitEvent(ms, e->ev_pmat, 0));
   }
   if (want_pipe && !skip_entry) {      // "everything on the main stream in front of this pruning kernel": what the next pipelined evaluation waits for
      HIPCHK(hipEventRecord(e->ev_entry[e->entry_sel], ms));
      e->entry_sel ^= 1;
      e->have_prev_entry = true;
   }
   e->n_pmat += (long)psets * (nn - 1);

   // Kernel B: fused pruning
   PruneArgs pr{};
   pr.ops = e->d_ops.p; pr.z = e->d_z.p; pr.z_stride = e->n_patt; pr.tiles = e->d_tiles.p; pr.n_tiles = e->n_tiles;
   pr.gene_off = e->d_gene_off.p; pr.weights = e->d_weights.p; pr.ztiles = e->d_ztiles.p; pr.zt_bytes = e->zt_bytes;
   pr.n = n; pr.n_tips = e->n_tips; pr.n_nodes = nn; pr.K = K; pr.n_genes = G; pr.n_codes = e->n_codes;
   pr.cleandata = e->cleandata; pr.n_pi = e->n_pi; pr.mode = e->mode; pr.n_scale = e->tree.n_scale;
   pr.keep = keep ? 1 : 0; pr.n_patt = e->n_patt;
   pr.pi = e->d_pi.p; pr.pint = e->kk == KK_MFMA64 ? e->d_pint.p : e->d_rowmajor.p; pr.ptip = e->d_ptip.p;
   if (e->use_jit && e->tree.n_scale) HIPCHK(e->d_fscale.ensure((size_t)K * e->n_patt));
   pr.fscale = e->d_fscale.p; pr.pcol = e->d_pcol.p;
   if (e->kk == KK_VALU20 && e->use_jit && e->m20) { pr.pint = e->d_pint.p; pr.pcol = e->d_rowmajor.p; }      // (operand-order P(t); the row-major copies for the all-4x4x4 experiment)
   pr.fhK = e->d_fhK.p; pr.partials = e->d_partials.p; pr.scalef = e->d_scalef.p; pr.stack_scratch = e->d_stack.p;
   pr.stack_overflow_slots = overflow; pr.first_matmul = e->prog.first_matmul; pr.n_int = n_int;
   pr.first_tip = e->prog.first_tip;
   pr.tile_group0 = e->d_tile_group0.p; pr.part_groups = e->part_groups();
   pr.part_dump = (keep && e->kk == KK_MFMA64) ? e->d_partials.p + (size_t)K * n_int * e->part_groups() * 1024 : nullptr;
   pr.code_mask = e->d_code_mask.p;
   pr.stream = e->d_stream.p; pr.n_stream = (int)(e->prog.stream.size() / 2); pr.tip_words = (long)tip_words(e);
   // the reduction's geometry (the fused kernels form the partial sums themselves; the others leave them to reduce_stage1)
   const int chunk = e->chunk, nbg = e->nb_global;
   const int nb = (e->n_patt + chunk - 1) / chunk;
   // PAML_AMD_OFFLOAD=1 (experiment): in a run of paml_amd_eval_device calls the whole reduction of evaluation i (mixture + log +
   // chunk sums, the exchange step, the fixed-order total) runs on the engine's side stream while the main stream goes straight on
   // to the pruning kernel of evaluation i + 1, two slots of class likelihoods alternating.  Measured on MI355X at the 8-GPU shard
   // size (profiles/r03_comm_overhead.txt): 0.2351 ms per evaluation against 0.2281 with the two small kernels left on the main
   // stream — the event record / wait pairs that order the streams cost what the kernel boundaries they remove did.  Not the default;
   // with a communicator only the all-reduce and the total go to the side stream.
   const bool fusedk = (e->kk != KK_MFMA64 && e->use_jit && e->fused) || coopj;      // the kernel forms the partial sums itself
   const bool offload = want_pipe && !fusedk && !keep && !clean && !bs && !want_lnf && !e->tree.n_scale && e->env.offload;
   const bool side_total = e->comm || dual;      // the (all-reduce and the) fixed-order total on the side stream `sc`
   if (!offload && !side_total)
      if (int rc = join_comm(e)) return rc;
   const int slot = (side_total || offload) ? e->red_slot : 0;
   if (e->comm_stats && !e->st_part[0])
      for (int i = 0; i < paml_amd_engine::NSTAT; i++) {
         HIPCHK(hipEventCreate(&e->st_part[i])); HIPCHK(hipEventCreate(&e->st_done[i]));
         HIPCHK(hipEventCreate(&e->st_w0[i])); HIPCHK(hipEventCreate(&e->st_w1[i]));
      }
   if (e->comm_stats) e->st_waited[e->st_count % paml_amd_engine::NSTAT] = false;
   DevBuf<double> &dpart = e->part_slot(slot);
   if ((size_t)nbg * B > dpart.cap) {
      if (e->sc) HIPCHK(hipStreamSynchronize(e->sc));      // (reallocation: nothing may still be reading the old buffer)
      HIPCHK(dpart.ensure((size_t)nbg * B));
      HIPCHK(hipMemsetAsync(dpart.p, 0, dpart.cap * sizeof(double), ms));
   }
   DevBuf<double> &dfhk = e->fhk_slot((offload || dual) ? slot : 0);
   if ((offload || dual) && slot) HIPCHK(dfhk.ensure((size_t)K * e->n_patt));
   pr.fhK = dfhk.p;
   e->last_fhk = (offload || dual) ? slot : 0;
   bool slot_waited = false;
   auto wait_slot = [&]() -> int {      // main stream: the reduction that last read this slot (two evaluations ago) is done
      if (slot_waited) return 0;
      slot_waited = true;
      if (e->done_pending[slot]) {
         e->done_pending[slot] = false;
         // (the caller's stream, once joined to that total by flush / enter, is already behind it; any other stream waits here)
         if (ms != e->stream || e->join_pending[slot]) {
            const int si = (int)(e->st_count % paml_amd_engine::NSTAT);
            if (e->comm_stats) HIPCHK(hipEventRecord(e->st_w0[si], ms));
            HIPCHK(hipStreamWaitEvent(ms, e->ev_done[slot], 0));
            if (e->comm_stats) { HIPCHK(hipEventRecord(e->st_w1[si], ms)); e->st_waited[si] = true; }
            if (ms == e->stream) e->join_pending[slot] = false;
         }
      }
      return 0;
   };
   if (offload) {
      if (int rc = ensure_side_stream(e)) return rc;
      if (int rc = wait_slot()) return rc;      // (the pruning kernel writes this slot's class likelihoods)
   }
   if ((size_t)B * RED_TICKET_WORDS > e->d_red_counter.cap) {
      HIPCHK(e->d_red_counter.ensure((size_t)std::max(B, 64) * RED_TICKET_WORDS));
      HIPCHK(hipMemsetAsync(e->d_red_counter.p, 0, e->d_red_counter.cap * sizeof(int), ms));
   }
   HIPCHK(e->d_out.ensure(B));
   if (want_lnf) HIPCHK(e->d_lnf.ensure((size_t)B * e->n_patt));
   double *const lnl_out = d_lnL_out ? d_lnL_out : e->d_out.p;
   const bool fused = fusedk;
   pr.zpm = e->d_zpm.p; pr.zpm_words = e->zpm_words;
   if (fused) {
      pr.zpm = e->d_zpm.p; pr.zpm_words = e->zpm_words; pr.Km = Km; pr.chunk = chunk; pr.first_chunk = e->first_chunk; pr.nb_stride = nbg;
      pr.want_fhk = (want_fhk || e->tree.n_scale) ? 1 : 0;
      pr.freqK = (bs && bs->freqK) ? e->d_b_freqK.p : e->d_freqK.p; pr.freqK_bs = (bs && bs->freqK) ? Km : 0;
      pr.lnf = want_lnf ? e->d_lnf.p : nullptr;
      pr.red_partial = dpart.p; pr.red_out = lnl_out; pr.nb_local = nb;
      if (int rc = wait_slot()) return rc;      // (this kernel writes the partial sums itself)
      // the total: a one-block stage-2 launch (default), or PAML_AMD_TAIL=1: the workgroup that finishes last forms it (tickets)
      pr.red_counter = (e->comm || !e->env.tail) ? nullptr : e->d_red_counter.p;
      if (coopj) pr.red_counter = e->d_red_counter.p;      // (the cooperative per-tree kernel: always the last workgroup, of each batch element)
   }
   const int prof_stride = std::max((int)e->prog.ops.size() + 3, e->env.prof_tiles ? 96 : 0);      // experiments only
   if (!e->env.prof_ops.empty()) {
      // per-op stamps: a fresh buffer and a dump after every launch; the workgroup timeline: one buffer, overwritten by every
      // launch and written out when the engine goes (nothing between the launches, so that the clock is the production clock)
      const size_t words = (size_t)3 * n_blocks * prof_stride;
      if (!e->env.prof_tiles || words != e->prof_words) {
         if (e->d_prof) (void)hipFree(e->d_prof);
         e->d_prof = nullptr;
         HIPCHK(hipMalloc((void **)&e->d_prof, paml_amd_engine::MAXL * words * 8));      // (pruning streams: one timeline per lane)
         HIPCHK(hipMemsetAsync(e->d_prof, 0, paml_amd_engine::MAXL * words * 8, ms));
         e->prof_words = words; e->prof_blocks = n_blocks; e->prof_stride = prof_stride;
      }
      pr.prof = e->d_prof + (size_t)lane * words;
      pr.prof_stride = prof_stride;
      pr.prof_tid = e->env.prof_tid;
   }
   mark(e);
   // CUs of the persistent kernels.  One pruning stream inside a communicator: a few stay free for the collective (engine_state.h,
   // comm_cus).  Runs of evaluations on two pruning streams: all of them — CUs left free by one kernel are taken at once by the
   // first workgroups of the next, so a reservation reserves nothing there and only costs tiles per CU (measured, 16 taxa x 10^6
   // codon patterns: 1.524-1.526 ms per evaluation on 256 CUs against 1.534-1.542 on 254; the same at the 8-GPU shard size, with
   // and without a communicator); the small kernels run when workgroups retire.
   const bool two_streams = want_pipe && e->env.dual && !e->profiling;
   const int cus = (e->comm && !two_streams) ? std::max(1, e->n_cu - e->comm_cus) : e->n_cu;
   switch (e->kk) {
   case KK_MFMA64:
      if (e->use_jit) {
         void *params[] = {&pr};
         const int grid = std::min(n_blocks, cus);     // persistent: one 130 KB-LDS workgroup per CU walks the tiles
         HIPCHK(hipModuleLaunchKernel(e->jit.fn, grid, 1, 1, e->mfma_waves * 64, 1, 1, 0, ms, params, nullptr));
      }
      else if (use_dma) {
         const size_t lds = (size_t)4 * 4096 * sizeof(double) + (size_t)e->n_tips * 128;
         if (!e->stream_attr_set) {
            HIPCHK(hipFuncSetAttribute((const void *)prune_mfma64_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            e->stream_attr_set = true;
         }
         hipLaunchKernelGGL(prune_mfma64_stream, dim3(n_blocks), dim3(512), lds, ms, pr);
      }
      else {
         // small data sets — every 16-pattern group can have a CU (ONE round: the kernel's 90 KB of LDS leave room for one workgroup
         // per CU, and two rounds of its 25 us walks lose to one round of the gather kernel's 39): four waves per group (prune_mfma64_coop)
         if (coopj) {
            void *params[] = {&pr};
            HIPCHK(hipModuleLaunchKernel(e->jit_coop.fn, e->n_tiles * 4 * K, 1, 1, 256, 1, 1, 0, ms, params, nullptr));
         }
         else if (coop) hipLaunchKernelGGL(prune_mfma64_coop, dim3(e->n_tiles * 4 * K), dim3(256), 0, ms, pr);
         else hipLaunchKernelGGL(prune_mfma64_gather<GATHER_WAVES>, dim3(n_blocks), dim3(GATHER_WAVES * 64), 0, ms, pr);
      }
      break;
   case KK_VALU4:
   case KK_VALU5:
   case KK_VALU20:
      if (fused) {
         void *params[] = {&pr};
         // single evaluations: one workgroup per chunk; batched ones: about two resident workgroups per CU in all, each walking every
         // gx-th chunk of its element (the LDS tables of an element's P(t) are filled once per workgroup, not once per 256 patterns)
         int gx = nb;
         if (B > 1 && !pr.red_counter && !e->fused_mfma4) gx = std::max(1, std::min(nb, 2 * e->n_cu / B));
         HIPCHK(hipModuleLaunchKernel(e->jit.fn, gx, B, 1, e->fused_threads, 1, 1, 0, ms, params, nullptr));
      }
      else if (e->use_jit && e->m20) {      // persistent: a multiple of the class count, every workgroup keeps its class's P(t) in LDS
         void *params[] = {&pr};
         int grid = std::min(std::max(K, cus / K * K), e->n_tiles * K);
         // Runs of evaluations on two pruning streams: the small kernels beside this one (partial sums, the next P(t)) are
         // dispatched to shader engines in turn, and a workgroup sent to an engine whose CUs all hold a persistent workgroup
         // waits for the end of this kernel even if the engine next door has room (measured: both then end WITH this kernel and
         // the next pruning kernel, which needs them, cannot be queued ahead).  7/8 of the CUs leave every shader engine one
         // free; taken when it costs this kernel nothing, i.e. when a wave still walks the same number of 32-pattern units
         // (10^5 patterns x 4 classes: 7 at 56 workgroups per class as at 63).  0.1885 -> 0.182 ms per evaluation (32 taxa).
         if (two_streams && G == 1) {
            const int units = std::min(e->n_tiles * 8, (e->n_patt + 31) / 32), g78 = e->n_cu * 7 / 8 / K * K;
            auto rounds = [&](int g) { return ((units + g / K - 1) / (g / K) + 7) / 8; };
            if (g78 >= K && g78 < grid && rounds(g78) == rounds(grid)) grid = g78;
         }
         static const int m20_threads = getenv("PAML_AMD_M20_W12") ? 768 : 512;      // (experiment: jit_generate_m20)
         // (several genes: a workgroup serves one (gene, class); the kernel deals a class's workgroups to the genes, at least one each)
         HIPCHK(hipModuleLaunchKernel(e->jit.fn, std::max(grid / K, G > 1 ? G : 1) * K, 1, 1, m20_threads, 1, 1, 0, ms, params, nullptr));
      }
      else if (e->use_jit) {
         void *params[] = {&pr};
         HIPCHK(hipModuleLaunchKernel(e->jit.fn, n_blocks, 1, 1, 256, 1, 1, 0, ms, params, nullptr));
      }
      else
         launch_valu(e, e->prog.max_stack, n_blocks, pr, ms);
      break;
   }
   mark(e);
   if (want_pipe && e->env.dual) {      // this P set's last reader so far
      HIPCHK(hipEventRecord(e->ev_setread[e->pset], ms));
      e->setread_rec[e->pset] = true;
   }
   if (pr.prof && !e->env.prof_tiles) {
      std::vector<unsigned long long> hp((size_t)3 * n_blocks * prof_stride);
      HIPCHK(hipMemcpyAsync(hp.data(), e->d_prof, hp.size() * 8, hipMemcpyDeviceToHost, ms));
      HIPCHK(hipStreamSynchronize(ms));
      FILE *f = fopen(e->env.prof_ops.c_str(), "wb");
      if (f) {
         int hdr[2] = {n_blocks, prof_stride};
         fwrite(hdr, sizeof(int), 2, f);
         std::vector<int> codes;
         for (auto &o : e->prog.ops) codes.push_back(o.code);
         codes.resize(prof_stride - 3, 0);
         fwrite(codes.data(), sizeof(int), codes.size(), f);
         fwrite(hp.data(), 8, hp.size(), f);
         fclose(f);
      }
   }

   // Kernel C: mixture + log + weighted sum.  Stage 1 leaves one partial sum per chunk of patterns at the chunk's global
   // position; with a communicator the ranks' (disjoint, zero elsewhere) arrays are summed over RCCL — adding zeros is exact,
   // so every rank then holds the same array whatever the number of ranks — and stage 2 adds it up in a fixed order.  On one
   // GPU the workgroup that finishes last forms the total itself (red_block_finish): no second launch.
   ReduceArgs ra{};
   ra.fhK = dfhk.p; ra.weights = e->d_weights.p; ra.freqK = e->d_freqK.p; ra.lnf = want_lnf ? e->d_lnf.p : nullptr;
   ra.partial = dpart.p; ra.out = lnl_out;
   ra.raw = ((e->kk == KK_MFMA64 && e->use_jit) || (e->kk == KK_VALU20 && e->use_jit && e->m20)) ? 1 : 0; ra.fscale = e->d_fscale.p;
   ra.n_patt = e->n_patt; ra.K = Km; ra.mode = e->mode; ra.n_scale = e->tree.n_scale; ra.chunk = chunk;
   ra.first_chunk = e->first_chunk; ra.nb_stride = nbg;
   // (measured on MI355X, 32 taxa x 10^5 nucleotide patterns: 28.2 us per evaluation with the separate one-block launch against
   //  30.2 with tickets — the agent-scope store + two atomics + coherent reads cross the XCDs' L2s and cost more than a launch)
   const bool tail = !side_total && !offload && e->env.tail;
   ra.counter = tail ? e->d_red_counter.p : nullptr;
   if (bs && bs->freqK) { ra.freqK = e->d_b_freqK.p; ra.freqK_bs = Km; }
   hipStream_t rs = ms;      // the stream of the reduction
   if (offload) {
      HIPCHK(hipEventRecord(e->ev_part[slot], ms));
      HIPCHK(hipStreamWaitEvent(e->sc, e->ev_part[slot], 0));
      rs = e->sc;
   }
   mark_on(e, rs);
   if (int rc = wait_slot()) return rc;
   if (!fused) hipLaunchKernelGGL(reduce_stage1, dim3(nb, B), dim3(256), 0, rs, ra);
   if (side_total) {
      // the exchange step, off the pruning stream: the side stream takes over when this evaluation's partial sums are there
      // (ev_part), all-reduces them into the slot's second buffer and forms the fixed-order total; the next evaluation's P(t) and
      // pruning kernel follow on the main stream without waiting for any of it
      DevBuf<double> &dtot = e->comm ? e->tot_slot(slot) : dpart;      // (one GPU, two pruning streams: only the total moves to `sc`)
      if (e->comm && (size_t)nbg * B > dtot.cap) {
         HIPCHK(hipStreamSynchronize(e->sc));
         HIPCHK(dtot.ensure((size_t)nbg * B));
      }
      if (!offload) {
         HIPCHK(hipEventRecord(e->ev_part[slot], ms));
         HIPCHK(hipStreamWaitEvent(e->sc, e->ev_part[slot], 0));
      }
      if (e->comm_stats) HIPCHK(hipEventRecord(e->st_part[e->st_count % paml_amd_engine::NSTAT], e->sc));      // (on `sc`, behind the wait: "partial sums ready")
      if (e->comm) {
         const ncclResult_t nr = rccl().AllReduce(dpart.p, dtot.p, (size_t)nbg * B, ncclDouble, ncclSum, e->comm, e->sc);
         if (nr != ncclSuccess) return fail(e, PAML_AMD_EHIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(nr));
      }
      hipLaunchKernelGGL(reduce_stage2, dim3(B), dim3(256), 0, e->sc, (const double *)dtot.p, nbg, ra.out);
   }
   else if (!tail && nbg > 1 && !coopj) hipLaunchKernelGGL(reduce_stage2, dim3(B), dim3(256), 0, rs, (const double *)dpart.p, nbg, ra.out);      // (one block per element: stage 1 wrote the total)
   if (side_total || offload) {
      if (e->comm_stats && side_total) { HIPCHK(hipEventRecord(e->st_done[e->st_count % paml_amd_engine::NSTAT], e->sc)); e->st_count++; }
      HIPCHK(hipEventRecord(e->ev_done[slot], e->sc));
      e->done_pending[slot] = true;
      e->join_pending[slot] = true;
      e->last_slot = slot;
      e->red_slot = (slot + 1) % e->n_lanes;
   }
   mark_on(e, rs);
   HIPCHK(hipGetLastError());
   if (e->profiling) e->prof_evals++;
   e->n_eval++;
   if (keep && !clean) e->partials_valid = true;
   e->pmat_valid = true;
   e->pipe_ok = want_pipe;      // (every other entry point clears it)
   // two pruning streams from the next call on: persistent kernels only (they are what holds every CU to its end), nothing shared
   // between consecutive evaluations but the two slots (class likelihoods, partial sums, P sets)
   e->dual_ok = want_pipe && e->env.dual && !e->env.offload && e->use_jit && (e->kk == KK_MFMA64 || (e->kk == KK_VALU20 && e->m20)) && !overflow &&
                !e->tree.n_scale && !keep && !clean && !bs && !want_lnf && nbg > 1 && (e->env.prof_ops.empty() || e->env.prof_tiles);
   return 0;
}

}  // namespace paml_amd

extern "C" {

int paml_amd_eval(paml_amd_engine *e, const double *branch, const double *gene_rate, double *lnL, double *lnf,
                  double *fhK)
{
   enter(e);
   if (!e || !branch || !lnL) return fail(e, PAML_AMD_EINVAL, "eval: null argument");
   int r = ensure_hout(e, 1);
   if (r) return r;
   r = launch_eval(e, branch, gene_rate, nullptr, e->h_out, lnf != nullptr, nullptr, false, fhK != nullptr);
   if (r) return r;
   if (lnf) HIPCHK(hipMemcpyAsync(lnf, e->d_lnf.p, (size_t)e->n_patt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (fhK)
      HIPCHK(hipMemcpyAsync(fhK, e->d_fhK.p, (size_t)e->K * e->n_patt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (int rc = join_comm(e)) return rc;
   HIPCHK(hipStreamSynchronize(e->stream));
   if (int rc = eigen_fail_check(e)) return rc;
   *lnL = e->h_out[0];
   return 0;
}

int paml_amd_eval_batch(paml_amd_engine *e, int n_batch, const double *branch, const double *gene_rate, const int *eigen_of,
                        const double *qfactor, const double *freqK, const double *rate, double *lnL, double *lnf)
{
   enter(e);
   if (!e || !branch || !lnL || n_batch < 1) return fail(e, PAML_AMD_EINVAL, "eval_batch: bad arguments");
   if ((long)n_batch * e->K * e->n_genes > 65535) return fail(e, PAML_AMD_EINVAL, "eval_batch: n_batch * K * n_genes > 65535");
   BatchSpec bs{n_batch, eigen_of, qfactor, freqK, rate};
   int r = ensure_hout(e, n_batch);
   if (r) return r;
   r = launch_eval(e, branch, gene_rate, nullptr, e->h_out, lnf != nullptr, &bs, false, false);
   if (r) return r;
   if (lnf)
      HIPCHK(hipMemcpyAsync(lnf, e->d_lnf.p, (size_t)n_batch * e->n_patt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (int rc = join_comm(e)) return rc;
   HIPCHK(hipStreamSynchronize(e->stream));
   if (int rc = eigen_fail_check(e)) return rc;
   memcpy(lnL, e->h_out, (size_t)n_batch * sizeof(double));
   return 0;
}

int paml_amd_eval_adg(paml_amd_engine *e, const double *branch, const double *gene_rate, const double *MK, const int *pose, int ls,
                      double *lnL)
{
   enter(e);
   if (!e || !branch || !MK || !pose || !lnL || ls < 1) return fail(e, PAML_AMD_EINVAL, "eval_adg: bad arguments");
   if (e->mode != PAML_AMD_MODE_LFUNDG) return fail(e, PAML_AMD_EINVAL, "eval_adg: needs the lfundG class mode");
   // With pattern shards pose[] holds GLOBAL pattern indices, the same on every rank: the class likelihoods are computed
   // on the shards, gathered over the ranks, and every rank runs the (sequential, short) chain over the sites itself.
   const bool sharded = e->comm && e->world > 1;
   const int K = e->K, npl = e->n_patt;
   const long np = sharded ? e->n_patt_global : npl;
   for (int i = 0; i < ls; i++)
      if (pose[i] < 0 || pose[i] >= np) return fail(e, PAML_AMD_EINVAL, "eval_adg: pose entry out of range");
   int r = launch_eval(e, branch, gene_rate, nullptr, nullptr, false);      // fx_r on the device
   if (r) return r;
   if (int rc = join_comm(e)) return rc;
   std::vector<double> fhK((size_t)K * np), w(np), b1(K), b2(K);
   const double *src = e->fhk_slot(e->last_fhk).p;
   if (sharded) {
      // the gather: every rank puts its columns into a zeroed [K + 1][n_patt_global] table (class likelihoods, then the weights) and
      // the tables are added — x + 0 is exact, so every rank ends with the bits a single engine would hold
      HIPCHK(e->d_adg_all.ensure((size_t)(K + 1) * np));
      HIPCHK(hipMemsetAsync(e->d_adg_all.p, 0, (size_t)(K + 1) * np * sizeof(double), e->stream));
      HIPCHK(hipMemcpy2DAsync(e->d_adg_all.p + e->first_patt, (size_t)np * sizeof(double), src, (size_t)npl * sizeof(double),
                              (size_t)npl * sizeof(double), (size_t)K, hipMemcpyDeviceToDevice, e->stream));
      HIPCHK(hipMemcpyAsync(e->d_adg_all.p + (size_t)K * np + e->first_patt, e->d_weights.p, (size_t)npl * sizeof(double),
                            hipMemcpyDeviceToDevice, e->stream));
      HIPCHK(hipEventRecord(e->ev_part[0], e->stream));
      HIPCHK(hipStreamWaitEvent(e->sc, e->ev_part[0], 0));
      const ncclResult_t nr = rccl().AllReduce(e->d_adg_all.p, e->d_adg_all.p, (size_t)(K + 1) * np, ncclDouble, ncclSum, e->comm, e->sc);
      if (nr != ncclSuccess) return fail(e, PAML_AMD_EHIP, std::string("eval_adg: ncclAllReduce: ") + rccl().GetErrorString(nr));
      HIPCHK(hipEventRecord(e->ev_done[0], e->sc));
      HIPCHK(hipStreamWaitEvent(e->stream, e->ev_done[0], 0));
      HIPCHK(hipMemcpyAsync(fhK.data(), e->d_adg_all.p, fhK.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(w.data(), e->d_adg_all.p + (size_t)K * np, w.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   } else {
      HIPCHK(hipMemcpyAsync(fhK.data(), src, fhK.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(w.data(), e->d_weights.p, w.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   }
   HIPCHK(hipStreamSynchronize(e->stream));
   if (int rc = eigen_fail_check(e)) return rc;
   // the chain over sites in their original order is sequential: host (treesub.c:7456-7492)
   double l = 0;
   if (e->tree.n_scale)
      for (long h = 0; h < np; h++) {
         const double fh = fhK[h];
         if (!(w[h] > 0)) continue;
         l += fh * w[h];
         fhK[h] = 1;
         for (int ir = 1; ir < K; ir++) fhK[(size_t)ir * np + h] = exp(fhK[(size_t)ir * np + h] - fh);
      }
   for (int il = 0; il < ls; il++) {
      const int h = pose[il];
      if (il == 0)
         for (int ir = 0; ir < K; ir++) b1[ir] = fhK[(size_t)ir * np + h];
      else {
         for (int ir = 0; ir < K; ir++) {
            double fh = 0;
            for (int j = 0; j < K; j++) fh += MK[ir * K + j] * b1[j];
            b2[ir] = fh * fhK[(size_t)ir * np + h];
         }
         b1 = b2;
      }
      double fh = 0;
      for (int ir = 0; ir < K; ir++) fh += b1[ir];
      if (fh < 1e-90) fh = 1e-300;
      for (int ir = 0; ir < K; ir++) b1[ir] /= fh;
      l += log(fh);
   }
   std::vector<double> fk(K);
   HIPCHK(hipMemcpy(fk.data(), e->d_freqK.p, K * sizeof(double), hipMemcpyDeviceToHost));
   double fh = 0;
   for (int ir = 0; ir < K; ir++) fh += fk[ir] * b1[ir];
   *lnL = l + log(fh);
   return 0;
}

int paml_amd_eval_device(paml_amd_engine *e, const double *branch, const double *gene_rate, double *d_lnL)
{
   if (!e || !branch || !d_lnL) return fail(e, PAML_AMD_EINVAL, "eval_device: null argument");
   return launch_eval(e, branch, gene_rate, nullptr, d_lnL, false, nullptr, true, false);
}

int paml_amd_flush(paml_amd_engine *e)
{
   if (!e) return PAML_AMD_EINVAL;
   if (int rc = join_comm(e)) return rc;
   // (no host synchronisation here: a decomposition that has ALREADY reported its sweep limit is returned now, one still running is
   //  caught by the next synchronous entry point — paml_amd_eigen_status waits for it)
   return eigen_fail_check(e);
}

int paml_amd_eigen_status(paml_amd_engine *e)
{
   if (!e) return PAML_AMD_EINVAL;
   if (int rc = join_comm(e)) return rc;
   HIPCHK(hipStreamSynchronize(e->stream));
   return eigen_fail_check(e);
}

int paml_amd_eval_dirty(paml_amd_engine *e, const double *branch, const double *gene_rate, const unsigned char *clean,
                        double *lnL)
{
   enter(e);
   if (!e || !branch || !lnL || !clean) return fail(e, PAML_AMD_EINVAL, "eval_dirty: null argument");
   int r = ensure_hout(e, 1);
   if (r) return r;
   r = launch_eval(e, branch, gene_rate, clean, e->h_out, false);
   if (r) return r;
   if (int rc = join_comm(e)) return rc;
   HIPCHK(hipStreamSynchronize(e->stream));
   if (int rc = eigen_fail_check(e)) return rc;
   *lnL = e->h_out[0];
   return 0;
}

}  // extern "C"
