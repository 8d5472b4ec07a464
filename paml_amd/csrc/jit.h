// jit.h — per-tree specialised pruning kernel.
//
// The interpreter kernels (kernels_prune.h) pay for their generality on every op: a switch dispatch, scalar
// loads of the op and stream tables, and — worst on CDNA4, where FP64 MFMA and VALU share the SIMD's
// issue — dozens of v_mov per op that the compiler needs to merge the loop-carried partials.  For a fixed
// tree the op sequence is known when paml_amd_set_tree returns, so this header unrolls it: it emits one
// straight-line HIP kernel (a few dozen calls into the hand-written building blocks of device_common.h,
// every block index, ring slot, wait count and register array a literal), compiles it for gfx950 with
// hiprtc and caches the module.  Partials are renamed instead of copied: each MFMA result is a fresh
// v4d[4] that *is* the next partial.  Same arithmetic, same operand stream, same LDS ring as
// prune_mfma64_stream; the interpreter remains the fallback (deep stacks, > 64 codes, > 128 tips, no hiprtc).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <cerrno>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "program.h"

namespace paml_amd {

struct JitKernel {
   hipModule_t mod = nullptr;
   hipFunction_t fn = nullptr;
   std::string key;
   size_t n_ops = 0;
   int stage = 0;            // large trees, two-stage build: which build this is (1 quick, 2 full; 0: compiled while the caller waited)
};

// Stack slots of the 61-state kernel that live in register arrays (each 32 VGPRs); deeper slots are spilled to global scratch.
// MFMA_RS (the interpreter's register slots) sizes the scratch: the engine allocates max_stack - MFMA_RS slots per workgroup.
constexpr int JIT_REG_SLOTS = 4;
constexpr int JIT_SCRATCH_BASE = 2;      // = MFMA_RS (checked in engine_state.h): scratch slot k holds stack slot k + 2

// Which programs the generator covers.
inline int jit_zpieces(int n_tips, int tp = 128) { return ((n_tips + 1) * tp + 2047) / 2048; }   // 2 KB units of a tile's code block (tp patterns per tile)

// LDS: ring, zb code blocks, pi, column tables, dump slot.  Two code blocks fit up to 95 tips, one up to 207.
inline bool jit_lds_fits(int n_tips, int zb, int tp = 128) { return 4 * 32768 + zb * jit_zpieces(n_tips, tp) * 2048 + 4 * 64 * 8 + (4 * 64 + 32) * 8 + 1024 <= 160 * 1024; }
inline int jit_zbuffers(int n_tips, int tp = 128) { return jit_lds_fits(n_tips, 2, tp) ? 2 : 1; }

// (STORE — every internal node's partial kept, PAML_AMD_KEEP_PARTIALS — is part of the tree's one program.  LOAD programs differ with the
//  set of clean nodes: a kernel per set would be compiled again and again, so they go to the interpreter unless `allow_load`.)
// Where a tile's tip codes live in LDS.  Up to 95 tips two blocks (the next tile's arrive while this one is walked), up to 207 one
// (replaced between tiles); beyond that the block is cut into PIECES by order of use: the rows (one per tip, then the weight flags) are
// laid out in the order the walk consumes them — every tip is read exactly once —, a piece is resident while the walk uses it and the
// next replaces it at the crossing (one wait for the DMA per tile and piece: nothing beside a tile's hundreds of products).  Round 5
// had two pieces (up to ~413 tips); round 6: as many as the tree needs (the reference's limit is NS 5000, codeml.c:19).
constexpr int JIT_ZP_MAX = 13;      // 2 KB units of LDS left for ONE code buffer beside the ring (27 392 bytes): 208 rows of 128 patterns
struct JitZPlan {
   int bufs = 2;             // LDS buffers (2: double-buffered; 1: one block or one piece)
   bool half = false;        // piece mode (the name is round 5's, when there were two)
   int pieces = 1;           // ... so many
   int ZP = 0;               // 2 KB units per buffer
   std::vector<int> row;     // tip (or n_tips: the flags) -> row in its piece
   std::vector<int> piece;   // tip (or n_tips) -> its piece
   std::vector<int> tip_of;  // row (over all pieces, in order of use) -> tip
   std::vector<int> row_at;  // row (over all pieces) -> row position in the tile's block: piece x (rows a buffer holds) + row in the piece
};
inline JitZPlan jit_zplan(const Program &p, int n_tips, int tp = 128)
{
   JitZPlan z;
   z.row.resize(n_tips + 1);
   z.piece.assign(n_tips + 1, 0);
   z.tip_of.resize(n_tips + 1);
   z.row_at.resize(n_tips + 1);
   for (int t = 0; t <= n_tips; t++) z.row[t] = z.tip_of[t] = z.row_at[t] = t;
   z.ZP = jit_zpieces(n_tips, tp);
   if (jit_lds_fits(n_tips, 2, tp)) { z.bufs = 2; return z; }
   z.bufs = 1;
   if (jit_lds_fits(n_tips, 1, tp)) return z;
   // pieces: rows in order of use, cut at op boundaries (the rows of one op — a cherry has two — stay together)
   z.half = true;
   std::vector<int> order;
   std::vector<char> seen(n_tips + 1, 0);
   std::vector<int> op_start;      // index into `order` where each tip-reading op starts
   for (const Op &o : p.ops) {
      const bool one = o.code == OP_SET_TIP || o.code == OP_MUL_TIP || o.code == OP_INIT_TIP, two = o.code == OP_SET_TIP2 || o.code == OP_MUL_TIP2;
      if (!one && !two) continue;
      op_start.push_back((int)order.size());
      if (!seen[o.a]) { seen[o.a] = 1; order.push_back(o.a); }
      if (two && !seen[o.b]) { seen[o.b] = 1; order.push_back(o.b); }
   }
   for (int t = 0; t < n_tips; t++)
      if (!seen[t]) order.push_back(t);      // (tips the program never reads: rows nobody looks at)
   order.push_back(n_tips);                  // the weight flags: read by ROOT, last
   const int total = n_tips + 1, cap = JIT_ZP_MAX * 2048 / tp;      // rows a buffer holds at most
   const int npc = std::max(2, (total + cap - 1) / cap);
   const int want = (total + npc - 1) / npc;      // even pieces (two: the halves of round 5), each cut moved up to the next op boundary
   std::vector<int> first(1, 0);      // first row of every piece
   while (total - first.back() > (first.size() == 1 && npc == 2 ? want : std::min(cap, std::max(want, 1)))) {
      int cut = first.back() + want;
      int at = -1;
      for (int st : op_start)
         if (st >= cut) { at = st; break; }
      if (at < 0 || at - first.back() > cap) {      // no boundary at or after the even cut within the buffer: the last one before it
         at = -1;
         for (int st : op_start)
            if (st > first.back() && st <= first.back() + cap) at = st;
         if (at < 0) at = std::min(total - 1, first.back() + cap);      // (cannot happen: an op has at most two rows)
      }
      if (at >= total) break;
      first.push_back(at);
   }
   first.push_back(total);
   z.pieces = (int)first.size() - 1;
   int rows_max = 0;
   for (int k = 0; k < z.pieces; k++) rows_max = std::max(rows_max, first[k + 1] - first[k]);
   z.ZP = (rows_max * tp + 2047) / 2048;
   const int rows_buf = z.ZP * 2048 / tp;
   for (int k = 0; k < z.pieces; k++)
      for (int r = first[k]; r < first[k + 1]; r++) {
         z.tip_of[r] = order[r];
         z.row[order[r]] = r - first[k];
         z.piece[order[r]] = k;
         z.row_at[r] = k * rows_buf + (r - first[k]);
      }
   return z;
}
inline bool jit_zfits(const Program &p, int n_tips, int tp = 128)
{
   const JitZPlan z = jit_zplan(p, n_tips, tp);
   return 4 * 32768 + z.bufs * z.ZP * 2048 + 4 * 64 * 8 + (4 * 64 + 32) * 8 + 1024 <= 160 * 1024;
}

inline bool jit_supported(const Program &p, int n_tips, int n_codes, int n_pi = 1, int max_arrays = 6, int tp = 128, bool allow_load = false)
{
   if (n_codes > 256 || p.ops.size() > 8000 || n_pi > 4) return false;      // (more than 64 codes: JIT_AMB_OVERFLOW, device_common.h)
   if (!jit_zfits(p, n_tips, tp)) return false;
   if (p.stream.size() / 2 < 4) return false;                       // trees this small go to the interpreter
   for (const Op &o : p.ops)
      if (o.code == OP_LOAD && !allow_load) return false;
   (void)max_arrays;      // stack slots beyond JIT_REG_SLOTS register arrays go to global scratch
   return true;
}

// 61 states: rows 48..60 of every product without the padded rows 61..63 (device_common.h, JitRowTail; pmat_kernel layout 3) —
// built and measured in round 4 (profiles/r04_61state.txt): bit-compatible results, 6 % fewer matrix-pipe cycles per product, and
// no faster (1.573 against 1.569 ms at 16 taxa x 10^6 patterns): the 45 extra operand fetches, the row-60 dot product and their waits
// take out of the shared issue port what the dropped rows put in.  Kept behind PAML_AMD_JIT_ROWTAIL=1; the default is the
// four-row-block form.
// Timing experiments whose kernels compute garbage (ablations: no barriers, no rank-1 seed, skewed waves, resident partials not stored,
// tip factors / operands left out): read only by a library built with -DPAML_AMD_JIT_EXPERIMENTS (PAML_AMD_EXTRA_FLAGS + engine.build(),
// tools/build_variant.sh); a production library ignores the variables.
inline const char *jit_experiment_env(const char *name)
{
#ifdef PAML_AMD_JIT_EXPERIMENTS
   return getenv(name);
#else
   (void)name;
   return nullptr;
#endif
}
inline bool jit_rowtail(int n_states) { return n_states == 61 && getenv("PAML_AMD_JIT_ROWTAIL") && !getenv("PAML_AMD_JIT_NOTAIL"); }

inline std::string jit_program_key(const Program &p, int n_tips)
{
   std::ostringstream k;
   k << n_tips << ":";
   for (const Op &o : p.ops) k << o.code << "," << o.a << "," << o.b << ";";
   return k.str();
}

// Emit the straight-line kernel for one program.
//
// Schedule (all static): the program's operand blocks are consumed in stream order through a ring of four LDS buffers
// that runs on across tiles (JIT2_* in device_common.h).  Every block step is  s_waitcnt vmcnt(N) + barrier  with N =
// the DMA pieces issued after the block, then the ring is refilled up to three blocks ahead — inside the MFMA loop of
// the step's matmul where there is one.  A cherry (SET_TIP2) that follows a pushed matmul is gathered under that
// matmul's second half; the first cherry of the NEXT tile is gathered under the current tile's last matmul (the tile's
// own first cherry was done that way by its predecessor; the first tile's is peeled in front of the loop).
// `first` = operand blocks of a tile already requested when the loop body starts (the body's last step leaves the
// same number of the next tile's in flight; *first_out reports it so that jit_generate can make the two agree).
// A tile's walk is one straight line of code — several hundred KB for a large tree — and the hardware runs such a line several times
// slower than the same instructions with a branch every few ops (230 tips, 256 tiles per launch: 5.8 ms against 1.03 ms; an `s_branch` to
// the next instruction does as well as a compiler-visible one: profiles/r05_big_tree_split.txt).  Programs of more than JIT_SPLIT_OPS ops
// get a never-taken uniform branch (JIT_SPLIT) every eighth op; the compiler then also works on blocks of a few thousand instructions
// instead of one of 10^5 (192 taxa: 28 s instead of 130 on this container's core).   0: none, 1: s_branch, 2: JIT_SPLIT
// Small programs get the compiler-visible form too: no front-end effect there (a bare s_branch changes nothing at 16 taxa), but the
// compiler's schedule of the shorter blocks measures 0.6 % faster (kernel 1.554 -> 1.544 ms at 16 taxa x 10^6 patterns, two runs each).
static const size_t JIT_SPLIT_OPS = 8;
inline int jit_split_mode(size_t nops)
{
   if (const char *v = getenv("PAML_AMD_JIT_SPLIT")) return !strcmp(v, "asm") ? 1 : !strcmp(v, "br") ? 2 : 0;
   return nops > JIT_SPLIT_OPS ? 2 : 0;
}
inline std::string jit_generate_impl(const Program &p, int n_tips, int n_states, int n_codes, int first, int *first_out, int waves = 8)
{
   std::ostringstream s;
   const int nblk = (int)p.stream.size() / 2;
   const int TP = waves * 16;
   const JitZPlan zpl = jit_zplan(p, n_tips, TP);
   const int ZP = zpl.ZP, ZR = (ZP * 8 + waves - 1) / waves;      // code block (or half): 2 KB units; DMA rounds per wave
   const size_t nops = p.ops.size();
   // states beyond n are zero padding: only RB row blocks and KB k-blocks of every P take part (4 and 16 at 61 states)
   const int RB = (n_states + 15) / 16, KB = (n_states + 3) / 4, KB2 = (KB + 1) / 2, NPc = KB2;   // NPc: 16-byte pieces per tip-table row
   const bool fuse_tips = !getenv("PAML_AMD_JIT_NOFUSE") && KB2 >= 2;   // cherries gathered under the preceding matmul
   const int MID = KB2 / 2;                                             // ... from this k-block pair on (jit_matvec_tip2)
   const bool spread = !getenv("PAML_AMD_JIT_NOSPREAD");    // ring refill issued from inside the MFMA loops
   const bool proft = getenv("PAML_AMD_PROF_TILES") != nullptr;            // kernel experiments: s_memrealtime (100 MHz) at workgroup start and at the end of each of its tiles
   const bool prof = !proft && getenv("PAML_AMD_PROF_OPS") != nullptr;      // kernel experiments: s_memtime stamp at every op
   // 61 states: the last k-block of P is the single column 60 — its rank-1 term goes through the vector pipe (a 512-byte
   // column table travels with every P block as a fifth DMA piece) and the k-block's four MFMAs are dropped
   const bool tail61 = n_states == 61 && !getenv("PAML_AMD_JIT_NOTAIL");
   int last_mm = -1;
   for (size_t i = 0; i < nops; i++)
      if (p.ops[i].code == OP_MATMUL || p.ops[i].code == OP_MATMUL_POP) last_mm = (int)i;
   // the next tile's first cherry rides under this tile's last matmul
   // (only when that matmul is the tile's last consumer of operand blocks: the ring has no room for more)
   bool tail_blocks = false;
   for (size_t i = last_mm + 1; i < nops; i++)
      if (p.ops[i].code == OP_SET_TIP || p.ops[i].code == OP_MUL_TIP || p.ops[i].code == OP_SET_TIP2 || p.ops[i].code == OP_MUL_TIP2)
         tail_blocks = true;
   // one code block only (large trees): it is replaced between tiles, so nothing of the next tile can start early
   const bool zsingle = zpl.bufs == 1, zhalf = zpl.half;
   int cur_piece = 0;         // (piece mode) the piece of the tile's codes that is in LDS
   const bool peel = fuse_tips && !getenv("PAML_AMD_JIT_NOPEEL") && nops > 2 && p.ops[0].code == OP_SET_TIP2 && last_mm > 1 &&
                     p.ops[1].code != OP_MUL_TIP && p.ops[1].code != OP_MUL_TIP2 && !tail_blocks && !zsingle;

   // chunks of a tip table that hold codes of this data set (two codes per 1 KB chunk)
   const int TCH = (n_codes + 1) / 2 >= 31 ? 32 : (n_codes + 1) / 2;
   // DMA rounds per block: a P has chunks pair * 4 + row block (< KB2 * 4 used), a tip table TCH chunks; `waves` chunks per round
   const int P_ROUNDS = (KB2 * 4 + waves - 1) / waves, T_ROUNDS = (TCH + waves - 1) / waves;
   s << "#define JIT_KB2 " << KB2 << "\n#define JIT_RB " << RB << "\n#define JIT_TCH " << TCH << "\n#define JIT_WAVES " << waves << "\n";
   if (jit_rowtail(n_states)) s << "#define JIT_ROWTAIL 1\n";
#ifdef TIP_SWZ_OFF
   s << "#define TIP_SWZ_OFF 1\n";      // (the library's P(t) kernel writes the tip tables without the swizzle: the per-tree kernel must read them so)
#endif
   // how STORE writes (device_common.h): 0 spread under the next product (default), 1 at the op; 2 — not at all, the ceiling measurement — is an experiment
   if (const char *v = getenv("PAML_AMD_JIT_STORE"))
      if (atoi(v) != 2 || jit_experiment_env("PAML_AMD_JIT_STORE")) s << "#define JIT_STORE_MODE " << atoi(v) << "\n";
   if (getenv("PAML_AMD_JIT_NT_STORE")) s << "#define JIT_NT_STORE 1\n";         // experiment: non-temporal stores of the class likelihoods
   if (jit_experiment_env("PAML_AMD_JIT_ABL_NOSEED")) s << "#define JIT_ABL_NOSEED 1\n";      // timing experiment: the rank-1 seed without its LDS reads and multiplies
   if (jit_experiment_env("PAML_AMD_JIT_ABL_NOBAR")) s << "#define JIT_ABL_NOBAR 1\n";      // timing experiment: no workgroup barriers (results are garbage)
   const char *abl_skew = jit_experiment_env("PAML_AMD_JIT_ABL_SKEW");                       // ... and waves 4-7 start this many x 64 cycles late
   if (zsingle) s << "#define JIT_ZB 1\n";
   if (zhalf) s << "#define JIT_ZPIECES " << zpl.pieces << "\n";
   const bool amb_over = n_codes > 64;      // codes beyond the 64 a ring block has rows for: summed from the rows of their states (device_common.h)
   if (amb_over) s << "#define JIT_AMB_OVERFLOW 1\n";
   const std::string ambarg = amb_over ? ", amb" : "";
   s << "#include \"device_common.h\"\nusing namespace paml_amd;\n";
   s << "extern \"C\" __global__ __launch_bounds__(" << waves * 64 << ", " << waves / 4 << ") void prune_jit(PruneArgs a)\n{\n";
   s << "   JIT2_PROLOGUE(" << ZP << ")\n";
   if (amb_over) {      // the state sets of the codes 64 .. : into the three quarters of sPi a single frequency vector leaves unused (192 x 8 bytes)
      s << "   if (a.n_pi == 1) for (int i = tid; i < a.n_codes - 64; i += JIT_WAVES * 64) ((unsigned long long *)sPi)[64 + i] = a.code_mask[64 + i];\n";
      s << "   const JitAmb amb{a.code_mask, (__attribute__((address_space(3))) const unsigned long long *)(sPi + 64), a.n_pi == 1};\n";
   }
   if (abl_skew) s << "   if (wave >= 4) { for (int i_ = 0; i_ < " << atoi(abl_skew) << "; i_++) __builtin_amdgcn_s_sleep(1); }\n";
   s << "   roff = " << ((4 - nblk % 4) & 3) << ";\n";

   // ---- static bookkeeping of what is in flight (per thread: pieces = vector-memory instructions) -------------------
   struct Item { int id, pieces; };          // id: block number within the tile (>= nblk: next tile's), -1: a code block
   std::vector<Item> fl;                     // issued, not yet known landed; oldest first
   int issued = 0, consumed = 0;
   auto n_rounds = [&](int blk) { return p.stream[2 * (blk % nblk)] ? T_ROUNDS : P_ROUNDS; };
   auto n_pieces = [&](int blk) { return n_rounds(blk) + ((tail61 && !p.stream[2 * (blk % nblk)]) ? 1 : 0); };
   auto piece = [&](int blk, int c4) {       // source text of one DMA piece of block blk (the last one of a P: the column table)
      const bool nx = blk >= nblk;
      const int loc = blk % nblk, is_tip = p.stream[2 * loc], node = p.stream[2 * loc + 1];
      if (c4 == n_rounds(blk)) return std::string("JIT2_PIECE_") + (nx ? "NPC(" : "PC(") + std::to_string(blk) + ", " + std::to_string(node) + ");";
      return std::string("JIT2_PIECE_") + (nx ? "N" : "") + (is_tip ? "T(" : "P(") + std::to_string(blk) + ", " + std::to_string(node) + ", " +
             std::to_string(c4) + ");";
   };
   auto issue_now = [&]() {
      s << "  ";
      for (int c4 = 0; c4 < n_pieces(issued); c4++) s << " " << piece(issued, c4);
      s << "\n";
      fl.push_back({issued, n_pieces(issued)});
      issued++;
   };
   auto wait_count = [&](int blk) {          // pieces that may stay in flight once block blk has to be complete
      int pos = -1;
      for (size_t i = 0; i < fl.size(); i++)
         if (fl[i].id == blk) pos = (int)i;
      if (pos < 0) return -1;                // already covered by an earlier wait
      int nfl = 0;
      for (size_t i = pos + 1; i < fl.size(); i++) nfl += fl[i].pieces;
      fl.erase(fl.begin(), fl.begin() + pos + 1);
      return nfl;
   };
   bool z_pending = false;                   // the next tile's code block still has to be requested in this tile
   std::vector<std::string> pend_store;      // the pieces of a STORE that ride under the product that follows it (see step)
   // make the next c blocks visible, then top the ring up — at once, or (defer) as a `side` functor that spreads the
   // refill's pieces over the first `iters` k-block pairs of the matmul that follows; the first `now` blocks from
   // `consumed` are needed within this very step and are never delayed
   auto step = [&](int c, bool defer = false, int iters = 8, int now = 1) -> std::string {
      // (a step whose blocks the ring has not been asked for yet — a cherry right after the tile's first one, as the forests of the branch-local
      //  refill have them: the top-up of the step before stopped short of it.  Every wave is done with the slots they go to after a barrier.)
      if (issued < consumed + c) {
         s << "   JIT_SYNC();\n";
         while (issued < consumed + c) issue_now();
      }
      const int nw = wait_count(consumed + c - 1);
      if (nw >= 0) s << "   JIT_WAIT(" << nw << ");";
      s << "   JIT_SYNC();\n";
      if (z_pending) {
         s << "   JIT2_ISSUE_Z(" << ZP << ")\n";
         fl.push_back({-1, ZR});
         z_pending = false;
      }
      const int upto = consumed + 4;
      if (!defer || !spread) {
         if (!pend_store.empty()) {
            s << "  ";
            for (const std::string &st : pend_store) s << " " << st;
            s << "\n";
            fl.push_back({-2, (int)pend_store.size()});
            pend_store.clear();
         }
         while (issued < upto) issue_now();
         return "JitNoSide()";
      }
      while (issued < consumed + now && issued < upto) issue_now();
      // a STORE waiting for this product (keep-partials mode: the partial it multiplies): its eight 1 KB wave stores go out one per
      // k-block pair, under the MFMAs — issued together they hold the CU's store path for ~5 000 cycles during which no wave of the
      // workgroup gets to its MFMAs (measured: 2.2 ms per evaluation at 16 taxa x 10^6 codon patterns against 1.56 without the stores).
      // The in-flight list below records them where they are issued, between the deferred DMA pieces of the same iterations.
      std::vector<std::string> stores;
      stores.swap(pend_store);
      struct Piece { int blk; std::string text; };      // blk = -2: a store
      std::vector<Piece> pieces;
      while (issued < upto) {
         for (int c4 = 0; c4 < n_pieces(issued); c4++) pieces.push_back({issued, piece(issued, c4)});
         issued++;
      }
      if (pieces.empty() && stores.empty()) return "JitNoSide()";
      const int per = pieces.empty() ? 1 : ((int)pieces.size() + iters - 1) / iters;
      std::vector<std::vector<Piece>> at(iters);
      for (size_t i = 0; i < pieces.size(); i++) at[i / per].push_back(pieces[i]);
      for (size_t i = 0; i < stores.size(); i++) at[i * iters / stores.size()].push_back({-2, stores[i]});
      // the in-flight list in the order of issue (a block whose pieces span iterations has several entries: wait_count takes the last)
      std::string f = "[&](int kb2) {";
      for (int i = 0; i < iters; i++) {
         if (at[i].empty()) continue;
         f += " if (kb2 == " + std::to_string(i) + ") {";
         for (const Piece &pc : at[i]) {
            f += " " + pc.text;
            if (!fl.empty() && fl.back().id == pc.blk) fl.back().pieces++;
            else fl.push_back({pc.blk, 1});
         }
         f += " }";
      }
      return f + " }";
   };
   const int prof_every = getenv("PAML_AMD_PROF_EVERY") ? std::max(1, atoi(getenv("PAML_AMD_PROF_EVERY"))) : 1;      // (experiments: a stamp at every n-th op only)
   const int split_mode = jit_split_mode(nops);      // (large trees: a branch every `split_every` ops, see jit_split_mode)
   const int split_every = getenv("PAML_AMD_JIT_SPLIT_EVERY") ? std::max(1, atoi(getenv("PAML_AMD_JIT_SPLIT_EVERY"))) : 8;
   auto stamp = [&](size_t iop) {
      if (prof && iop % prof_every == 0)
         s << "   if (a.prof && tid == a.prof_tid && ptile) a.prof[(long)blockIdx.x * a.prof_stride + 1 + " << iop
           << "] = __builtin_amdgcn_s_memtime();\n";
      else if (!prof && split_mode && iop && iop % split_every == 0)
         s << (split_mode == 1 ? "   asm volatile(\"s_branch 0\");\n" : "   JIT_SPLIT()\n");
   };
   auto code = [&](int tip) { return "JIT2_CODE(" + std::to_string(ZP) + ", " + std::to_string(zpl.row[tip]) + ")"; };
   auto ncode = [&](int tip) { return "JIT2_NCODE(" + std::to_string(ZP) + ", " + std::to_string(zpl.row[tip]) + ")"; };
   const std::string issue_z = zhalf ? "JIT2_ISSUE_ZH(" + std::to_string(ZP) + ", n_tile, 0)" : "JIT2_ISSUE_Z(" + std::to_string(ZP) + ")";
   // (piece mode) rows of a later piece are needed from here on: every wave is done with the piece in LDS, the next comes over it
   auto cross_if = [&](std::initializer_list<int> tips) {
      if (!zhalf) return;
      int need = cur_piece;
      for (int t : tips) need = std::max(need, zpl.piece[t]);
      while (cur_piece < need) {
         cur_piece++;
         s << "   __syncthreads();\n   JIT2_ISSUE_ZH(" << ZP << ", cur_tile, " << cur_piece << ")\n   JIT_WAIT(0); __syncthreads();\n";
         fl.clear();
      }
   };
   auto buf = [&](int blk) { return "JIT2_BUF(" + std::to_string(blk) + ")"; };
   auto colarg = [&](int blk) { return tail61 ? ", JIT2_COL(" + std::to_string(blk) + "), x60" : std::string(); };

   // ---- in front of the loop: the first tile is the "next" tile of an empty predecessor ---------------------------
   const int reg_slots = std::min(p.max_stack, JIT_REG_SLOTS);
   const int NA = reg_slots + 2 + (fuse_tips ? 1 : 0);
   const int SPILLED = -2;
   auto spill_ptr = [&](int slot_no) { return "JIT_SPILL_PTR(" + std::to_string(slot_no - JIT_SCRATCH_BASE) + ")"; };
   for (int i = 0; i < NA; i++) s << "   v4d A" << i << "[4];\n";
   if (peel) s << "   v4d AS[4];\n";       // the first cherry of a tile, produced under the predecessor's last matmul
   s << "   JIT2_NEXT_SET()\n   " << issue_z << "\n";
   fl.push_back({-1, ZR});
   issued = nblk;                           // numbered as the blocks after the (empty) predecessor's
   for (int i = 0; i < first; i++) issue_now();
   if (peel) {
      const int nw = wait_count(nblk + 1);
      s << "   JIT_WAIT(" << nw << "); __syncthreads();\n";
      s << "   jit_tip2_set<" << NPc << ">(AS, " << buf(nblk) << ", " << ncode(p.ops[0].a) << ", " << buf(nblk + 1) << ", " << ncode(p.ops[0].b) << ", q, lane" << ambarg << ");\n";
   }
   if (zsingle) {      // everything requested so far has to be there when the loop starts: the same state the loop's end leaves
      s << "   JIT_WAIT(0); __syncthreads();\n";
      fl.clear();
   }
   // renumber for the loop body: those three blocks are blocks 0..2 of the tile the loop starts with
   for (Item &it : fl)
      if (it.id >= 0) it.id -= nblk;
   issued = first;
   consumed = peel ? 2 : 0;

   if (proft) s << "   int ptc = 0; if (a.prof && tid == 0) { a.prof[(long)blockIdx.x * a.prof_stride] = __builtin_amdgcn_s_memrealtime(); a.prof[(long)blockIdx.x * a.prof_stride + a.prof_stride - 2] = __builtin_amdgcn_s_memtime(); }\n";
   bool resident = false;      // the program stores or loads resident partials (keep-partials mode)
   for (const Op &o : p.ops) resident = resident || o.code == OP_STORE || o.code == OP_LOAD;
   s << "   int ptile = 1;\n   for (;; ptile = 0) {\n";
   s << "   JIT2_ADVANCE(" << nblk << ")\n";
   // (n_tile is still this tile's number here; the groups of a tile's waves that start past its gene's end belong to nobody: not stored)
   if (resident) s << "   const int tg0 = as_const(a.tile_group0)[n_tile];\n   const bool wave_in = h0 + wave * 16 < hend;\n";
   if (zhalf) s << "   const int cur_tile = n_tile;\n";
   s << "   work += gridDim.x;\n   JIT2_NEXT_SET()\n";
   z_pending = !zsingle;

   // register arrays: a free list; `cur` names the array holding the partial under construction
   std::vector<int> freeA;
   for (int i = NA - 1; i >= 0; i--) freeA.push_back(i);
   const int AS = 1000;
   auto alloc = [&]() { int r = freeA.back(); freeA.pop_back(); return r; };
   auto release = [&](int r) { if (r != AS) freeA.push_back(r); };
   auto name = [&](int r) { return r == AS ? std::string("AS") : "A" + std::to_string(r); };
   std::vector<int> slot(256, -1);   // stack slot -> array
   int cur = peel ? AS : -1;

   if (prof) s << "   if (a.prof && tid == a.prof_tid && ptile) a.prof[(long)blockIdx.x * a.prof_stride] = __builtin_amdgcn_s_memtime();\n";
   // An op that starts a partial while one is still held: a forest of subtrees (the branch-local refill, engine_branch.hip) — the previous
   // subtree's root was stored and is done.  (Not written over in place: after a first subtree that is a lone cherry the array is AS, which the
   // tile's last product fills for the NEXT tile while this tile's later partials would still live in it.)
   auto start_partial = [&]() {
      if (cur >= 0) release(cur);
      cur = alloc();
   };
   for (size_t iop = 0; iop < nops; iop++) {
      const Op &o = p.ops[iop];
      stamp(iop);
      if (iop == 0 && peel) continue;      // done by the predecessor
      switch (o.code) {
      case OP_INIT_ONES:
         start_partial();
         s << "   jit_init_ones(" << name(cur) << ", q, n);\n";
         break;
      case OP_INIT_TIP:
         start_partial();
         cross_if({o.a});
         s << "   jit_init_tip(" << name(cur) << ", " << code(o.a) << ", q, a.cleandata);\n";
         break;
      case OP_SET_TIP:
         start_partial();
         cross_if({o.a});
         step(1);
         s << "   jit_tip_set<" << NPc << ">(" << name(cur) << ", " << buf(consumed) << ", " << code(o.a) << ", q, lane" << ambarg << ");\n";
         consumed += 1;
         break;
      case OP_MUL_TIP:
         cross_if({o.a});
         step(1);
         s << "   jit_tip_mul<" << NPc << ">(" << name(cur) << ", " << buf(consumed) << ", " << code(o.a) << ", q, lane" << ambarg << ");\n";
         consumed += 1;
         break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2:
         if (o.code == OP_SET_TIP2 || cur < 0) start_partial();
         cross_if({o.a, o.b});
         step(2);
         s << "   " << (o.code == OP_SET_TIP2 ? "jit_tip2_set<" : "jit_tip2_mul<") << NPc << ">(" << name(cur) << ", " << buf(consumed) << ", " << code(o.a)
           << ", " << buf(consumed + 1) << ", " << code(o.b) << ", q, lane" << ambarg << ");\n";
         consumed += 2;
         break;
      case OP_PUSH:
         if (o.b >= JIT_REG_SLOTS) {
            s << "   jit_spill(" << name(cur) << ", " << spill_ptr(o.b) << ");\n";
            release(cur);
            slot[o.b] = SPILLED;
         }
         else
            slot[o.b] = cur;
         cur = -1;
         break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         const int pop = mm_pop_slot(o), push = mm_push_slot(o);
         const int out = alloc();
         // a cherry right after a pushed matmul: its two tip gathers ride under this matmul's second half
         // (piece mode: a cherry whose codes lie in the NEXT piece is not folded under this product — the crossing then happens at the cherry's
         //  own step; there are at most pieces - 1 of them per tile.  The folded form is kept behind PAML_AMD_JIT_FOLD_CROSSING=1: one run of the
         //  1 000-tip tree of tests/test_engine_gpu.py came out wrong with it during development and could not be reproduced — the identical
         //  generated source gives the oracle's value on the final build — so the form every full test run has exercised is the default.)
         const bool crossing_cherry = zhalf && iop + 1 < nops && p.ops[iop + 1].code == OP_SET_TIP2 &&
                                      std::max(zpl.piece[p.ops[iop + 1].a], zpl.piece[p.ops[iop + 1].b]) > cur_piece;
         const bool fuse = fuse_tips && push >= 0 && iop + 1 < nops && p.ops[iop + 1].code == OP_SET_TIP2 &&
                           (!crossing_cherry || getenv("PAML_AMD_JIT_FOLD_CROSSING"));
         const bool fuse_next = peel && (int)iop == last_mm;     // ... or the next tile's first cherry under the last matmul
         if (fuse) cross_if({p.ops[iop + 1].a, p.ops[iop + 1].b});
         // (tip tables the ring could not hold earlier are requested in the first k-block pairs and awaited at the midpoint)
         // (the cross-lane read of x[60] is issued before the step's wait + barrier so that its latency hides there)
         if (tail61) s << "   { const double x60 = jit_x60(" << name(cur) << ", lane);\n";
         const std::string side = step(1, true, (fuse || fuse_next) ? MID : KB2, 1);
         int tgt = -1;
         if (fuse || fuse_next) {
            const Op &nx = fuse ? p.ops[iop + 1] : p.ops[0];
            tgt = fuse ? alloc() : AS;
            const int mid = wait_count(consumed + 2);
            s << "   jit_matvec_tip2<" << (mid < 0 ? 63 : mid) << (tail61 ? ", true, " : ", false, ") << RB << ", " << KB << ">(" << buf(consumed) << ", lane, " << name(cur)
              << ", " << name(out) << ", " << buf(consumed + 1) << ", " << (fuse ? code(nx.a) : ncode(nx.a)) << ", " << buf(consumed + 2) << ", "
              << (fuse ? code(nx.b) : ncode(nx.b)) << ", q, " << name(tgt) << ", " << side << colarg(consumed) << (amb_over ? (tail61 ? ", amb" : ", nullptr, 0.0, amb") : "") << ");"
              << (tail61 ? " }" : "") << "\n";
            consumed += 3;
         }
         else {
            s << "   jit_matvec<" << (tail61 ? "true" : "false") << ", " << RB << ", " << KB << ">(" << buf(consumed) << ", lane, " << name(cur) << ", "
              << name(out) << ", "
              << side << colarg(consumed) << ");" << (tail61 ? " }" : "") << "\n";
            consumed += 1;
         }
         release(cur);
         if (pop >= 0) {
            if (slot[pop] == SPILLED)
               s << "   jit_mul_mem(" << name(out) << ", " << spill_ptr(pop) << ");\n";
            else {
               s << "   jit_mul(" << name(out) << ", " << name(slot[pop]) << ");\n";
               release(slot[pop]);
            }
            slot[pop] = -1;
         }
         if (push >= JIT_REG_SLOTS) {
            s << "   jit_spill(" << name(out) << ", " << spill_ptr(push) << ");\n";
            release(out);
            slot[push] = SPILLED;
            cur = -1;
         }
         else if (push >= 0) {
            slot[push] = out;
            cur = -1;
         }
         else
            cur = out;
         if (fuse) {      // the SET_TIP2 is done
            cur = tgt;
            iop++;
            stamp(iop);
         }
      } break;
      case OP_SCALE:
         s << "   { const double fac = jit_scale(" << name(cur) << ", q, n); lnscale += fac;\n"
           << "     if (a.keep && q == 0 && valid) a.scalef[((long)iclass * a.n_scale + " << o.b << ") * a.n_patt + h] = fac; }\n";
         break;
      // STORE / LOAD: eight vector-memory operations per wave, entered in the in-flight list like the ring's DMA pieces, so that the
      // counted waits that follow let them fly (a wait computed without them would make every block step wait for the stores'
      // completion in HBM).  At a tile's start the list lacks the previous tile's last stores: those waits are stricter than needed, never laxer.
      case OP_STORE: {
         const bool under_product = spread && iop + 1 < nops && (p.ops[iop + 1].code == OP_MATMUL || p.ops[iop + 1].code == OP_MATMUL_POP) && p.ops[iop + 1].a == o.a;
         if (under_product)      // (the product that follows reads this very array and leaves it alone: see step)
            for (int i = 0; i < 8; i++)
               pend_store.push_back("JIT_STORE_PIECE(" + name(cur) + ", JIT_PART_DST(" + std::to_string(o.a) + "), " + std::to_string(i) + ");");
         else {
            s << "   jit_store(" << name(cur) << ", JIT_PART_DST(" << o.a << "), lane);\n";
            fl.push_back({-2, 8});
         }
      } break;
      case OP_LOAD:
         start_partial();
         s << "   jit_load(" << name(cur) << ", JIT_PART_PTR(" << o.a << "), lane);\n";
         fl.push_back({-2, 8});
         break;
      case OP_ROOT:
         // keep-partials mode with scaling nodes: the factors of clean subtrees were stored by earlier evaluations — all of them are
         // summed from memory in slot order, as the interpreter kernels do (MFMA_ROOT_CASE; treesub.c:7746-7747)
         cross_if({n_tips});      // (the weight flags: the last row)
         if (resident)
            s << "   if (a.keep && a.n_scale) { lnscale = 0; if (valid) for (int k_ = 0; k_ < a.n_scale; k_++) lnscale += a.scalef[((long)iclass * a.n_scale + k_) * a.n_patt + h]; }\n";
         s << "   jit_root_lds(a, " << name(cur) << ", lnscale, sPi + (a.n_pi > 1 ? gene : 0) * 64, " << code(n_tips)
           << ", iclass, q, h, valid);\n";
         release(cur);
         cur = -1;
         break;
      default: break;
      }
   }
   // programs whose steps never came by a barrier after the tile switch (no operand blocks): request the codes here
   if (z_pending) {
      s << "   __syncthreads();\n   JIT2_ISSUE_Z(" << ZP << ")\n";
      fl.push_back({-1, ZR});
   }
   if (zsingle) {      // all waves are done with this tile's codes: fetch the next tile's (first half) over them, and wait (once per tile)
      s << "   __syncthreads();\n   " << issue_z << "\n   JIT_WAIT(0); __syncthreads();\n";
      fl.clear();
   }
   if (proft) s << "   if (a.prof && tid == 0 && ptc < a.prof_stride - 4) { a.prof[(long)blockIdx.x * a.prof_stride + 1 + ptc] = __builtin_amdgcn_s_memrealtime(); a.prof[(long)blockIdx.x * a.prof_stride + a.prof_stride - 1] = __builtin_amdgcn_s_memtime(); }\n   ptc++;\n";
   s << "   if (!has_next) break;\n   }\n   JIT_WAIT(0);\n}\n";
   *first_out = issued - nblk;
   return s.str();
}

inline std::string jit_generate(const Program &p, int n_tips, int n_states = 61, int n_codes = 64, int waves = 8)
{
   int first = 3, got = 3;
   std::string src = jit_generate_impl(p, n_tips, n_states, n_codes, first, &got, waves);
   if (got != first) {
      first = got;
      src = jit_generate_impl(p, n_tips, n_states, n_codes, first, &got, waves);
   }
   if (got != first) return std::string("#error \"jit schedule does not close\"\n");
   // (one basic block of > 120 ops: the quick build first.  With the block splits the full passes take no longer than the quick ones.)
   if (p.ops.size() > 120 && jit_split_mode(p.ops.size()) != 2) src = "// JIT_BIG: compiled with JIT_BIG_FLAGS (jit_compile_code)\n" + src;
   return src;
}

// The one-pattern-per-lane kernels (4 / 5 / 20 states) specialised the same way: the op interpreter of prune_valu<N>
// unrolled into straight-line code over renamed register arrays.
inline bool jit_valu_supported(const Program &p, int max_arrays = 8)
{
   if (p.ops.size() > 600) return false;
   for (const Op &o : p.ops)
      if (o.code == OP_STORE || o.code == OP_LOAD || o.code == OP_EXPORT) return false;
   return p.max_stack + 2 <= max_arrays;
}

inline std::string jit_generate_valu(const Program &p, int N)
{
   std::ostringstream s;
   s << "#include \"device_common.h\"\nusing namespace paml_amd;\n";
   s << "extern \"C\" __global__ __launch_bounds__(256) void prune_jit(PruneArgs a)\n{\n   JV_PROLOGUE(" << N << ")\n";
   const int NA = p.max_stack + 2;
   for (int i = 0; i < NA; i++) s << "   double A" << i << "[N];\n";
   std::vector<int> freeA;
   for (int i = NA - 1; i >= 0; i--) freeA.push_back(i);
   auto alloc = [&]() { int r = freeA.back(); freeA.pop_back(); return r; };
   auto release = [&](int r) { freeA.push_back(r); };
   auto name = [&](int r) { return "A" + std::to_string(r); };
   std::vector<int> slot(256, -1);
   int cur = -1;
   const char *LOOP = "_Pragma(\"unroll\") for (int j = 0; j < N; j++) ";
   for (const Op &o : p.ops) {
      switch (o.code) {
      case OP_INIT_ONES:
         if (cur < 0) cur = alloc();
         s << "   " << LOOP << name(cur) << "[j] = 1.0;\n";
         break;
      case OP_INIT_TIP:
         if (cur < 0) cur = alloc();
         s << "   { const int c = JV_CODE(" << o.a << "); " << LOOP << name(cur) << "[j] = (a.cleandata && j == c) ? 1.0 : 0.0; }\n";
         break;
      case OP_SET_TIP:
      case OP_MUL_TIP:
         if (cur < 0) cur = alloc();
         s << "   { const double *r = JV_ROW(" << o.a << ", JV_CODE(" << o.a << ")); " << LOOP << name(cur)
           << (o.code == OP_SET_TIP ? "[j] = r[j]; }\n" : "[j] *= r[j]; }\n");
         break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2:
         if (cur < 0) cur = alloc();
         s << "   { const double *r1 = JV_ROW(" << o.a << ", JV_CODE(" << o.a << ")), *r2 = JV_ROW(" << o.b << ", JV_CODE(" << o.b << ")); "
           << LOOP << name(cur) << (o.code == OP_SET_TIP2 ? "[j] = r1[j] * r2[j]; }\n" : "[j] = (" + name(cur) + "[j] * r1[j]) * r2[j]; }\n");
         break;
      case OP_PUSH:
         slot[o.b] = cur;
         cur = -1;
         break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         const int pop = mm_pop_slot(o), push = mm_push_slot(o), out = alloc();
         s << "   jv_matvec<N>(Pint + " << (long)o.a * N * N << ", " << name(cur) << ", " << name(out) << ");\n";
         release(cur);
         if (pop >= 0) {
            s << "   " << LOOP << name(out) << "[j] = " << name(slot[pop]) << "[j] * " << name(out) << "[j];\n";
            release(slot[pop]);
            slot[pop] = -1;
         }
         if (push >= 0) { slot[push] = out; cur = -1; }
         else cur = out;
      } break;
      case OP_SCALE:
         s << "   { const double fac = jv_scale<N>(" << name(cur) << "); lnscale += fac;\n"
           << "     if (a.keep && valid) a.scalef[((long)iclass * a.n_scale + " << o.b << ") * a.n_patt + h] = fac; }\n";
         break;
      case OP_ROOT:
         s << "   jv_root<N>(a, " << name(cur) << ", lnscale, gene, iclass, h, valid);\n";
         release(cur);
         cur = -1;
         break;
      default: break;
      }
   }
   s << "}\n";
   return s.str();
}

// ---- fused variant (4 / 5 states): classes as the inner loop, tip factors from LDS tables, reduction in the epilogue --------
// Specialised on the program, the number of states N, of character codes NC, of classes K and on the shape below.
// See the block comment at jvf_row_set (device_common.h) for the design.
//   R  = patterns per lane (1 or 2): with two, every P(t) entry fetched through the scalar cache feeds two FMAs and the
//        waits on those fetches come half as often — the big-problem shape;
//   CW = class groups per workgroup (1, 2 or 4): the workgroup has 256 x CW threads, group g walks the classes g, g + CW, ...
//        of the chunk's 256 patterns and the mixture is summed (in class order) by group 0 from an LDS exchange — the
//        small-problem shape: a data set of 10^5 patterns is only ~1.5 waves per SIMD otherwise.
struct ValuFusedPlan {
   bool ok = false;
   bool cherry = false;      // product tables for SET_TIP2 cherries
   int n_cherry = 0, R = 1, CW = 1;
   size_t lds_bytes = 0;
};

inline ValuFusedPlan jit_valu_fused_plan(const Program &p, int N, int n_tips, int n_codes, int K, int chunk)
{
   ValuFusedPlan pl;
   if (!jit_valu_supported(p) || (N != 4 && N != 5) || n_tips > 255 || K < 1) return pl;
   for (const Op &o : p.ops)
      if (o.code == OP_SET_TIP2) pl.n_cherry++;
   // shape: a function of the (global) chunk size only, so that every rank of a sharded evaluation picks the same kernel
   // (measured on MI355X, 32 taxa x Gamma-4: two class groups beat one and four at 10^5 and at 4 x 10^6 patterns — more waves to
   //  hide the scalar-cache latency of the P(t) fetches than one group, a smaller P(t) working set than four; two patterns per
   //  lane cost more in registers than they save: profiles/r02_valu_fused_shapes.txt)
   (void)chunk;
   pl.CW = K >= 2 ? 2 : 1;
   pl.R = 1;
   if (const char *v = getenv("PAML_AMD_VF_R")) pl.R = atoi(v) == 2 && chunk >= 512 ? 2 : 1;
   if (const char *v = getenv("PAML_AMD_VF_CW")) pl.CW = std::max(1, std::min(std::min(4, K), atoi(v)));
   if (pl.CW == 3) pl.CW = 2;
   if (pl.CW > 1) pl.R = 1;
   const size_t rows = (size_t)K * n_tips * n_codes * N * 8;
   const size_t ch = (size_t)K * pl.n_cherry * n_codes * n_codes * N * 8;
   const size_t xch = pl.CW > 1 ? (size_t)K * 256 * 8 : 0;
   if (rows + xch > 60 * 1024) return pl;               // (the unfused kernel, gathering from L2, takes such models)
   pl.ok = true;
   pl.cherry = pl.n_cherry > 0 && n_codes <= 8 && rows + ch + xch <= 56 * 1024 && !getenv("PAML_AMD_VF_NOCHERRY");
   pl.lds_bytes = rows + (pl.cherry ? ch : 0) + xch;
   return pl;
}

// Several genes (option G: com.posG / com.rgene / com.piG; G > 1 here): a gene has its own P(t) — gene rate, per Mgene also its own
// eigen system and frequencies — so the LDS tables are one gene's.  A chunk is walked gene segment by gene segment: the tables are
// refilled where the gene changes (a workgroup's chunks ascend, so at most G + 1 times), and the 256-pattern sub-tile a boundary falls
// into is walked once per gene with the other gene's lanes switched off — every pattern keeps the lane and the turn it has in
// reduce_stage1's order, so the chunk sums keep their bits.  With G == 1 the generated text is the single-gene kernel's, unchanged.
inline std::string jit_generate_valu_fused(const Program &p, int N, int n_tips, int n_codes, int K, int chunk, int G = 1)
{
   const ValuFusedPlan pl = jit_valu_fused_plan(p, N, n_tips, n_codes, K, chunk);
   const bool MG = G > 1;
   std::ostringstream s;
   const int R = MG ? 1 : pl.R, CW = pl.CW;
   const int NC = n_codes, ROWW = n_tips * NC * N, CHW = pl.cherry ? pl.n_cherry * NC * NC * N : 0, TABW = ROWW + CHW;   // doubles per class
   const int ZW = ((n_tips + 3) / 4 + 3) / 4 * 4;      // dwords of packed codes per pattern
   const int NTH = 256 * CW;
   s << "#include \"device_common.h\"\nusing namespace paml_amd;\n";
   // the LDS tables: the tips are nodes 0 .. NT-1, so a class's rows are one contiguous run of pmat's output; for a cherry the products of
   // its two tips' rows.  Several genes: the fill is needed inside the loop over the chunks, where — inlined — the compiler computes its
   // thirty-odd loop-invariant addresses once, in front of that loop, and keeps them in registers through the whole walk (142 VGPRs
   // instead of 111: one workgroup per CU instead of two); it is a function of its own there, called where the gene changes.
   std::ostringstream fill;      // the body, in terms of `ptip0` (the tables of the element's class 0) and `cstride` (doubles per class)
   {
      fill << "   for (int ir = 0; ir < K; ir++) {\n"
           << "      const double *src = ptip0 + ir * cstride;\n"
           << "      for (int i = threadIdx.x; i < ROWW; i += NTH) sTab[ir * TABW + i] = src[i];\n"
           << "   }\n   __syncthreads();\n";
      if (pl.cherry) {
         int c = 0;
         fill << "   for (int i = threadIdx.x; i < K * NC * NC * N; i += NTH) {\n"
              << "      const int ir = i / (NC * NC * N), r = i % (NC * NC * N), ca = r / (NC * N), cb = (r / N) % NC, j = r % N;\n"
              << "      const double *rw = sTab + ir * TABW;\n";
         for (const Op &o : p.ops)
            if (o.code == OP_SET_TIP2) {
               fill << "      sTab[ir * TABW + ROWW + " << c * NC * NC * N << " + r] = rw[(" << o.a << " * NC + ca) * N + j] * rw[(" << o.b << " * NC + cb) * N + j];\n";
               c++;
            }
         fill << "   }\n   __syncthreads();\n";
      }
   }
   const std::string consts = "constexpr int N = " + std::to_string(N) + ", NC = " + std::to_string(NC) + ", K = " + std::to_string(K) + ", NT = " + std::to_string(n_tips) +
                              ", ROWW = " + std::to_string(ROWW) + ", TABW = " + std::to_string(TABW) + ", ZW = " + std::to_string(ZW) + ", CW = " + std::to_string(CW) +
                              ", NTH = " + std::to_string(NTH) + ";\n";
   if (MG) {
      s << consts << "__shared__ __attribute__((aligned(16))) double sTab[K * TABW];\n";
      s << "__device__ __attribute__((noinline)) void vf_fill(const double *ptip0, long cstride)\n{\n" << fill.str() << "}\n";
   }
   s << "extern \"C\" __global__ __launch_bounds__(" << NTH << (R > 1 ? ", 2" : "") << ") void prune_jit(PruneArgs a)\n{\n";
   if (!MG) {
      s << "   " << consts << "   (void)NT;\n";
      s << "   __shared__ __attribute__((aligned(16))) double sTab[K * TABW];\n";
   }
   if (CW > 1) s << "   __shared__ double sF[K * 256];\n";
   s << "   const int tid = threadIdx.x & 255, cw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), bat = blockIdx.y;\n   (void)cw;\n";
   s << "   const long cls0 = (long)bat * K;\n";
   auto emit_fill = [&](const char *ind, const char *pset0) {      // pset0: parameter set of the batch element's class 0 (gene x all classes + cls0)
      if (MG) s << ind << "vf_fill(a.ptip + (" << pset0 << ") * a.n_nodes * a.tip_words, (long)a.n_nodes * a.tip_words);\n";
      else s << "   { const double *ptip0 = a.ptip + (" << pset0 << ") * a.n_nodes * a.tip_words; const long cstride = (long)a.n_nodes * a.tip_words;\n" << fill.str() << "   }\n";
   };
   if (!MG) emit_fill("   ", "cls0");
   s << "   const CONST_AS double *fK = as_const(a.freqK + bat * a.freqK_bs);\n";
   if (!MG) s << "   const CONST_AS double *pi = as_const(a.pi);\n";
   else {
      // the G + 1 gene offsets sit in scalar registers for the whole launch (independent loads, one round trip): finding a chunk's
      // gene is a chain of scalar compares — a search through memory cost every chunk a string of dependent scalar loads with
      // nothing to overlap them (measured: the batched 4-gene launch 14 % slower than the one-gene one, a single evaluation 13 us)
      s << "   int tab_gene = -1;\n   const CONST_AS int *goff = as_const(a.gene_off);\n";
      for (int g = 0; g <= G; g++) s << "   const long go" << g << " = " << "goff[" << g << "];\n";
   }
   // the workgroup's chunks: its own (gridDim.x = number of chunks: single evaluations), or every gridDim.x-th one (batched
   // evaluations: fewer, longer workgroups per element, so that the tables above are filled once per ~100 chunks)
   s << "   for (int cb = blockIdx.x; cb < a.nb_local; cb += gridDim.x) {\n";
   s << "   const long c_lo = (long)cb * a.chunk, c_hi = (c_lo + a.chunk < (long)a.n_patt) ? c_lo + a.chunk : (long)a.n_patt;\n";
   s << "   double acc = 0;\n";
   if (MG) {
      s << "   int gene = 0; long g_lo = go0, g_hi = go1;\n";
      for (int g = 1; g < G; g++) s << "   if (c_lo >= go" << g << ") { gene = " << g << "; g_lo = go" << g << "; g_hi = go" << g + 1 << "; }\n";
      s << "   for (;;) {\n";
      s << "   const long seg_lo = g_lo > c_lo ? g_lo : c_lo, seg_hi = g_hi < c_hi ? g_hi : c_hi;\n";
      s << "   if (seg_lo < seg_hi) {\n";
      s << "   const long gset0 = (long)gene * a.K + cls0;\n";
      s << "   const CONST_AS double *pi = as_const(a.pi + (a.n_pi > 1 ? gene : 0) * N);\n";
      s << "   for (long h0 = c_lo + (seg_lo - c_lo) / 256 * 256; h0 < seg_hi; h0 += 256) {\n";
   }
   else s << "   for (long h0 = c_lo; h0 < c_hi; h0 += " << 256 * R << ") {\n";
   auto sfx = [&](int r) { return R > 1 ? "_" + std::to_string(r) : std::string(); };
   for (int r = 0; r < R; r++) {
      const std::string x = sfx(r);
      if (MG)
         s << "      const long h" << x << " = h0 + " << 256 * r << " + tid;\n      const bool valid" << x << " = h" << x << " >= seg_lo && h" << x << " < seg_hi;\n      const long hc" << x
           << " = h" << x << " < c_hi ? h" << x << " : c_hi - 1;\n";      // (any pattern of the chunk: its codes are loaded before the gene's tables are known to be there)
      else
      s << "      const long h" << x << " = h0 + " << 256 * r << " + tid;\n      const bool valid" << x << " = h" << x << " < c_hi;\n      const long hc" << x
        << " = valid" << x << " ? h" << x << " : c_hi - 1;\n";
      s << "      unsigned int zw" << x << "[ZW];\n";
      s << "      { const uint4 *zp = (const uint4 *)(a.zpm + hc" << x << " * ZW);\n";
      for (int i = 0; i < ZW / 4; i++)
         s << "        { const uint4 t = zp[" << i << "]; zw" << x << "[" << 4 * i << "] = t.x; zw" << x << "[" << 4 * i + 1 << "] = t.y; zw" << x << "[" << 4 * i + 2
           << "] = t.z; zw" << x << "[" << 4 * i + 3 << "] = t.w; }\n";
      s << "      }\n";
      s << "      const double wt" << x << " = a.weights[hc" << x << "];\n";
      if (MG) {      // the tables of this segment's gene (the codes and the weight above are already on their way)
         s << "      if (gene != tab_gene) {\n         __syncthreads();\n";
         emit_fill("         ", "gset0");
         s << "         tab_gene = gene;\n      }\n";
      }
      // per-pattern table offsets (in doubles), formed once and used by every class
      s << "#define zw zw" << x << "\n";
      int c = 0;
      std::vector<char> seen(n_tips, 0);
      for (const Op &o : p.ops)
         if (o.code == OP_SET_TIP2 && pl.cherry) {
            s << "      const int oc" << c << x << " = ROWW + " << c * NC * NC * N << " + (JVF_CODE(" << o.a << ") * NC + JVF_CODE(" << o.b << ")) * N;\n";
            seen[o.a] = seen[o.b] = 1;
            c++;
         }
      for (const Op &o : p.ops) {
         auto tipoff = [&](int t) {
            if (!seen[t]) { s << "      const int ot" << t << x << " = (" << t << " * NC + JVF_CODE(" << t << ")) * N;\n"; seen[t] = 2; }
         };
         switch (o.code) {
         case OP_SET_TIP: case OP_MUL_TIP: tipoff(o.a); break;
         case OP_MUL_TIP2: tipoff(o.a); tipoff(o.b); break;
         case OP_SET_TIP2: if (!pl.cherry) { tipoff(o.a); tipoff(o.b); } break;
         case OP_INIT_TIP: s << "      const int ci" << o.a << x << " = JVF_CODE(" << o.a << ");\n"; break;
         default: break;
         }
      }
      s << "#undef zw\n";
      s << "      double fh" << x << " = 0, v" << x << " = 0;\n";
   }
   s << "      _Pragma(\"unroll 1\") for (int ir = " << (CW > 1 ? "cw" : "0") << "; ir < K; ir += CW) {\n";
   s << "         const double *Pint = a.pint + (" << (MG ? "gset0" : "cls0") << " + ir) * a.n_nodes * (N * N);\n";
   s << "         const double *tab = sTab + ir * TABW;\n";
   const int NA = p.max_stack + 2;
   for (int r = 0; r < R; r++) {
      s << "         double lnscale" << sfx(r) << " = 0;\n         (void)lnscale" << sfx(r) << ";\n";
      for (int i = 0; i < NA; i++) s << "         double A" << i << sfx(r) << "[N];\n";
   }
   std::vector<int> freeA;
   for (int i = NA - 1; i >= 0; i--) freeA.push_back(i);
   auto alloc = [&]() { int r = freeA.back(); freeA.pop_back(); return r; };
   auto release = [&](int r) { freeA.push_back(r); };
   auto name = [&](int a, int r) { return "A" + std::to_string(a) + sfx(r); };
   std::vector<int> slot(256, -1);
   int cur = -1, ich = 0;
   const char *LOOP = "_Pragma(\"unroll\") for (int j = 0; j < N; j++) ";
   for (const Op &o : p.ops) {
      int out = -1, pop = -1, push = -1, curin = cur;
      if (o.code == OP_INIT_ONES || o.code == OP_INIT_TIP || o.code == OP_SET_TIP || o.code == OP_SET_TIP2) {
         if (cur < 0) cur = alloc();
         curin = cur;
      }
      if (o.code == OP_MATMUL || o.code == OP_MATMUL_POP) { pop = mm_pop_slot(o); push = mm_push_slot(o); out = alloc(); }
      for (int r = 0; r < R; r++) {
         const std::string x = sfx(r), C = name(curin, r);
         switch (o.code) {
         case OP_INIT_ONES: s << "         " << LOOP << C << "[j] = 1.0;\n"; break;
         case OP_INIT_TIP: s << "         " << LOOP << C << "[j] = (a.cleandata && j == ci" << o.a << x << ") ? 1.0 : 0.0;\n"; break;
         case OP_SET_TIP: s << "         jvf_row_set<N>(" << C << ", tab + ot" << o.a << x << ");\n"; break;
         case OP_MUL_TIP: s << "         jvf_row_mul<N>(" << C << ", tab + ot" << o.a << x << ");\n"; break;
         case OP_SET_TIP2:
            if (pl.cherry) s << "         jvf_row_set<N>(" << C << ", tab + oc" << ich << x << ");\n";
            else
               s << "         { const double *r1 = tab + ot" << o.a << x << ", *r2 = tab + ot" << o.b << x << "; " << LOOP << C << "[j] = r1[j] * r2[j]; }\n";
            break;
         case OP_MUL_TIP2:
            s << "         { const double *r1 = tab + ot" << o.a << x << ", *r2 = tab + ot" << o.b << x << "; " << LOOP << C << "[j] = (" << C << "[j] * r1[j]) * r2[j]; }\n";
            break;
         case OP_MATMUL:
         case OP_MATMUL_POP:
            s << "         jv_matvec<N>(Pint + " << (long)o.a * N * N << ", " << C << ", " << name(out, r) << ");\n";
            if (pop >= 0) s << "         " << LOOP << name(out, r) << "[j] = " << name(slot[pop], r) << "[j] * " << name(out, r) << "[j];\n";
            break;
         case OP_SCALE: s << "         lnscale" << x << " += jv_scale<N>(" << C << ");\n"; break;
         case OP_ROOT:
            // fx_r treesub.c:7728-7749 / lfun 7780-7798, then this class's term of lfundG's mixture (7632-7652)
            s << "         { double f = 0;\n            " << LOOP << "f = fma(pi[j], " << C << "[j], f);\n"
              << "            if (a.mode == PAML_AMD_MODE_LFUN) { if (f <= 0) f = 1e-80; v" << x << " = log(f) + lnscale" << x << "; if (a.want_fhk && valid" << x
              << ") a.fhK[(cls0 + ir) * a.n_patt + h" << x << "] = wt" << x << " > 0 ? v" << x << " : 0.0; }\n"
              << "            else {\n"
              << "               if (f <= 0) f = 1e-300;\n"
              << "               if (a.n_scale) { if (valid" << x << ") a.fhK[(cls0 + ir) * a.n_patt + h" << x << "] = wt" << x << " > 0 ? log(f) + lnscale" << x << " : 0.0; }\n"
              << "               else {\n";
            if (CW > 1) s << "                  sF[ir * 256 + tid] = f;\n";
            else s << "                  fh" << x << " += fK[ir] * f;\n";
            s << "                  if (a.want_fhk && valid" << x << ") a.fhK[(cls0 + ir) * a.n_patt + h" << x << "] = wt" << x << " > 0 ? f : 0.0; }\n"
              << "            } }\n";
            break;
         default: break;
         }
      }
      switch (o.code) {
      case OP_SET_TIP2: if (pl.cherry) ich++; break;
      case OP_PUSH: slot[o.b] = cur; cur = -1; break;
      case OP_MATMUL:
      case OP_MATMUL_POP:
         release(curin);
         if (pop >= 0) { release(slot[pop]); slot[pop] = -1; }
         if (push >= 0) { slot[push] = out; cur = -1; }
         else cur = out;
         break;
      case OP_ROOT: release(cur); cur = -1; break;
      default: break;
      }
   }
   s << "      }\n";      // classes
   if (CW > 1) s << "      __syncthreads();\n";
   for (int r = 0; r < R; r++) {
      const std::string x = sfx(r);
      s << "      if (a.mode != PAML_AMD_MODE_LFUN" << (CW > 1 ? " && cw == 0" : "") << ") {\n"
           "         if (a.n_scale) {      /* log-sum-exp around the first maximum (treesub.c:7640-7649) */\n"
           "            const double *fk = a.fhK + cls0 * a.n_patt + hc" << x << ";\n"
           "            int it = 0;\n"
           "            for (int ir = 1; ir < K; ir++) if (fk[(long)ir * a.n_patt] > fk[(long)it * a.n_patt]) it = ir;\n"
           "            const double t = fk[(long)it * a.n_patt];\n"
           "            double fh = 0;\n"
           "            for (int ir = 0; ir < K; ir++) fh += fK[ir] * exp(fk[(long)ir * a.n_patt] - t);\n"
           "            v" << x << " = t + log(fh);\n"
           "         }\n"
           "         else {\n";
      if (CW > 1) s << "            for (int ir = 0; ir < K; ir++) fh" << x << " += fK[ir] * sF[ir * 256 + tid];\n";
      s << "            if (fh" << x << " <= 0) fh" << x << " = 1e-300;\n            v" << x << " = log(fh" << x << ");\n         }\n      }\n";
      s << "      if (!(valid" << x << " && wt" << x << " > 0)) v" << x << " = 0;\n";
      s << "      if (valid" << x << (CW > 1 ? " && cw == 0" : "") << ") { acc += v" << x << " * wt" << x << "; if (a.lnf) a.lnf[(long)bat * a.n_patt + h" << x << "] = v" << x << "; }\n";
   }
   if (CW > 1) s << "      __syncthreads();\n";      // sF is reused by the next sub-tile
   s << "   }\n";      // sub-tiles
   if (MG) {      // on to the chunk's next gene segment
      s << "   }\n   if (g_hi >= c_hi || gene + 1 >= " << G << ") break;\n   gene++; g_lo = g_hi;\n   g_hi = ";
      for (int g = 1; g < G - 1; g++) s << "gene == " << g << " ? go" << g + 1 << " : ";
      s << "go" << G << ";\n   }\n";
   }
   if (CW > 1) s << "   if (cw > 0) acc = 0;\n";
   s << "   red_block_finish<" << CW << ">(acc, a.red_partial + (long)bat * a.nb_stride, a.first_chunk + cb, a.nb_stride, a.red_out + bat, a.red_counter ? a.red_counter + bat * RED_TICKET_WORDS : nullptr);\n";
   s << "   }\n}\n";
   return s.str();
}

// ---- 4-state models on the matrix cores: v_mfma_f64_4x4x4_4b_f64 (jit_generate_mfma4) -------------------------------------------
// Measured on MI355X (profiles/r02_mfma4_layout.txt): the instruction runs at the full FP64 rate with a single wave per SIMD, and
// its result layout equals its B-operand layout — lane = 16 i + 4 b + j holds row i of column (b, j).  So a partial is ONE double
// per lane: state = lane >> 4, pattern = lane & 15 (four blocks of four patterns, all with the same A = P), and L' = P . L is one
// instruction per 16 patterns whose result is the next B operand.  P arrives as a per-lane vector operand (lane (k, b, i) holds
// P[i][k], a 128-byte line of the row-major matrix): no scalar-cache traffic, no lgkmcnt coupling with the LDS tip tables, which
// is what held the one-pattern-per-lane kernel at 0.51 - 0.63 of the FP64 peak.  Everything else is the fused kernel's frame:
// tip (and cherry) tables in LDS, classes inside, mixture + log + weighted chunk sum in the epilogue, CW class groups.
// A wave owns 64 patterns as G = 4 independent groups of 16 (four MFMA chains in flight); elementwise products, tip factors and
// the root stage use all 64 lanes (4 states x 16 patterns).
inline std::string jit_generate_mfma4(const Program &p, int n_tips, int n_codes, int K, int chunk)
{
   const ValuFusedPlan pl = jit_valu_fused_plan(p, 4, n_tips, n_codes, K, chunk);
   std::ostringstream s;
   const int N = 4, G = 4, CW = pl.CW;
   const int NC = n_codes, ROWW = n_tips * NC * N, CHW = pl.cherry ? pl.n_cherry * NC * NC * N : 0, TABW = ROWW + CHW;   // doubles per class
   const int ZW = ((n_tips + 3) / 4 + 3) / 4 * 4;
   const int NTH = 256 * CW;
   s << "#include \"device_common.h\"\nusing namespace paml_amd;\n";
   s << "extern \"C\" __global__ __launch_bounds__(" << NTH << ") void prune_jit(PruneArgs a)\n{\n";
   s << "   constexpr int N = 4, NC = " << NC << ", K = " << K << ", ROWW = " << ROWW << ", TABW = " << TABW << ", ZW = " << ZW << ", CW = " << CW
     << ", NTH = " << NTH << ";\n";
   s << "   __shared__ __attribute__((aligned(16))) double sTab[K * TABW];\n   __shared__ double sV[256];\n";
   if (CW > 1) s << "   __shared__ double sF[K * 256];\n";
   s << "   const int tid = threadIdx.x & 255, cw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), bat = blockIdx.y;\n   (void)cw;\n";
   s << "   const int lane = tid & 63, wv = tid >> 6, st = lane >> 4, col = lane & 15;\n";
   s << "   const long cls0 = (long)bat * K;\n";
   s << "   for (int ir = 0; ir < K; ir++) {\n"
        "      const double *src = a.ptip + (cls0 + ir) * a.n_nodes * a.tip_words;\n"
        "      for (int i = threadIdx.x; i < ROWW; i += NTH) sTab[ir * TABW + i] = src[i];\n"
        "   }\n   __syncthreads();\n";
   if (pl.cherry) {
      int c = 0;
      s << "   for (int i = threadIdx.x; i < K * NC * NC * N; i += NTH) {\n"
           "      const int ir = i / (NC * NC * N), r = i % (NC * NC * N), ca = r / (NC * N), cb = (r / N) % NC, j = r % N;\n"
           "      const double *rw = sTab + ir * TABW;\n";
      for (const Op &o : p.ops)
         if (o.code == OP_SET_TIP2) {
            s << "      sTab[ir * TABW + ROWW + " << c * NC * NC * N << " + r] = rw[(" << o.a << " * NC + ca) * N + j] * rw[(" << o.b << " * NC + cb) * N + j];\n";
            c++;
         }
      s << "   }\n   __syncthreads();\n";
   }
   s << "   const long c_lo = (long)blockIdx.x * a.chunk, c_hi = (c_lo + a.chunk < (long)a.n_patt) ? c_lo + a.chunk : (long)a.n_patt;\n";
   s << "   const CONST_AS double *fK = as_const(a.freqK + bat * a.freqK_bs);\n";
   s << "   const double pis = a.pi[st];\n";
   s << "   const int aidx = (lane & 3) * 4 + (lane >> 4);      /* A operand: lane (k, b, i) <- P[i][k] */\n";
   s << "   double acc = 0;\n";
   s << "   for (long h0 = c_lo; h0 < c_hi; h0 += 256) {\n";
   auto gx = [&](int g) { return "_" + std::to_string(g); };
   for (int g = 0; g < G; g++) {
      const std::string x = gx(g);
      s << "      const long h" << x << " = h0 + wv * 64 + " << 16 * g << " + col;\n      const bool valid" << x << " = h" << x << " < c_hi;\n";
      s << "      unsigned int zw" << x << "[ZW];\n      { const uint4 *zp = (const uint4 *)(a.zpm + (valid" << x << " ? h" << x << " : c_hi - 1) * ZW);\n";
      for (int i = 0; i < ZW / 4; i++)
         s << "        { const uint4 t = zp[" << i << "]; zw" << x << "[" << 4 * i << "] = t.x; zw" << x << "[" << 4 * i + 1 << "] = t.y; zw" << x << "[" << 4 * i + 2
           << "] = t.z; zw" << x << "[" << 4 * i + 3 << "] = t.w; }\n";
      s << "      }\n";
      s << "#define zw zw" << x << "\n";
      int c = 0;
      std::vector<char> seen(n_tips, 0);
      for (const Op &o : p.ops)
         if (o.code == OP_SET_TIP2 && pl.cherry) {
            s << "      const int oc" << c << x << " = ROWW + " << c * NC * NC * N << " + (JVF_CODE(" << o.a << ") * NC + JVF_CODE(" << o.b << ")) * N + st;\n";
            seen[o.a] = seen[o.b] = 1;
            c++;
         }
      for (const Op &o : p.ops) {
         auto tipoff = [&](int t) {
            if (!seen[t]) { s << "      const int ot" << t << x << " = (" << t << " * NC + JVF_CODE(" << t << ")) * N + st;\n"; seen[t] = 2; }
         };
         switch (o.code) {
         case OP_SET_TIP: case OP_MUL_TIP: tipoff(o.a); break;
         case OP_MUL_TIP2: tipoff(o.a); tipoff(o.b); break;
         case OP_SET_TIP2: if (!pl.cherry) { tipoff(o.a); tipoff(o.b); } break;
         case OP_INIT_TIP: s << "      const int ci" << o.a << x << " = JVF_CODE(" << o.a << ");\n"; break;
         default: break;
         }
      }
      s << "#undef zw\n";
      s << "      double fh" << x << " = 0, v" << x << " = 0;\n";
   }
   s << "      _Pragma(\"unroll 1\") for (int ir = " << (CW > 1 ? "cw" : "0") << "; ir < K; ir += CW) {\n";
   s << "         const double *Pint = a.pint + (cls0 + ir) * a.n_nodes * (N * N) + aidx;\n";
   s << "         const double *tab = sTab + ir * TABW;\n";
   const int NA = p.max_stack + 2;
   for (int g = 0; g < G; g++) {
      s << "         double lnscale" << gx(g) << " = 0;\n         (void)lnscale" << gx(g) << ";\n";
      s << "         double";
      for (int i = 0; i < NA; i++) s << (i ? ", " : " ") << "A" << i << gx(g) << " = 0";
      s << ";\n";
   }
   std::vector<int> freeA;
   for (int i = NA - 1; i >= 0; i--) freeA.push_back(i);
   auto alloc = [&]() { int r = freeA.back(); freeA.pop_back(); return r; };
   auto release = [&](int r) { freeA.push_back(r); };
   auto name = [&](int a, int g) { return "A" + std::to_string(a) + gx(g); };
   std::vector<int> slot(256, -1);
   int cur = -1, ich = 0;
   for (const Op &o : p.ops) {
      int out = -1, pop = -1, push = -1, curin = cur;
      if (o.code == OP_INIT_ONES || o.code == OP_INIT_TIP || o.code == OP_SET_TIP || o.code == OP_SET_TIP2) {
         if (cur < 0) cur = alloc();
         curin = cur;
      }
      if (o.code == OP_MATMUL || o.code == OP_MATMUL_POP) {
         pop = mm_pop_slot(o); push = mm_push_slot(o); out = alloc();
         s << "         { const double pa = Pint[" << (long)o.a * N * N << "];\n";
      }
      for (int g = 0; g < G; g++) {
         const std::string x = gx(g), C = name(curin, g);
         switch (o.code) {
         case OP_INIT_ONES: s << "         " << C << " = 1.0;\n"; break;
         case OP_INIT_TIP: s << "         " << C << " = (a.cleandata && st == ci" << o.a << x << ") ? 1.0 : 0.0;\n"; break;
         case OP_SET_TIP: s << "         " << C << " = tab[ot" << o.a << x << "];\n"; break;
         case OP_MUL_TIP: s << "         " << C << " *= tab[ot" << o.a << x << "];\n"; break;
         case OP_SET_TIP2:
            if (pl.cherry) s << "         " << C << " = tab[oc" << ich << x << "];\n";
            else s << "         " << C << " = tab[ot" << o.a << x << "] * tab[ot" << o.b << x << "];\n";
            break;
         case OP_MUL_TIP2: s << "         " << C << " = (" << C << " * tab[ot" << o.a << x << "]) * tab[ot" << o.b << x << "];\n"; break;
         case OP_MATMUL:
         case OP_MATMUL_POP:
            s << "           " << name(out, g) << " = __builtin_amdgcn_mfma_f64_4x4x4f64(pa, " << C << ", 0.0, 0, 0, 0);\n";
            break;
         case OP_SCALE:      // NodeScale treesub.c:7200-7230: the maximum over the four states sits on lane bits 4-5
            s << "         { double mx = " << C << " > 0 ? " << C << " : 0; { const double o1 = __shfl_xor(mx, 16); mx = o1 > mx ? o1 : mx; } { const double o2 = __shfl_xor(mx, 32); mx = o2 > mx ? o2 : mx; }\n"
              << "           if (mx < 1e-300) { " << C << " = 1.0; lnscale" << x << " += -800; } else { " << C << " /= mx; lnscale" << x << " += log(mx); } }\n";
            break;
         case OP_ROOT:
            s << "         { double f = pis * " << C << "; f += __shfl_xor(f, 16); f += __shfl_xor(f, 32);\n"
              << "            const bool own = st == 0 && valid" << x << ";\n"
              << "            const double wtz = own ? a.weights[h" << x << "] : 0.0;\n"
              << "            if (a.mode == PAML_AMD_MODE_LFUN) { if (f <= 0) f = 1e-80; v" << x << " = log(f) + lnscale" << x << "; if (a.want_fhk && own) a.fhK[(cls0 + ir) * a.n_patt + h" << x
              << "] = wtz > 0 ? v" << x << " : 0.0; }\n"
              << "            else {\n"
              << "               if (f <= 0) f = 1e-300;\n"
              << "               if (a.n_scale) { if (own) a.fhK[(cls0 + ir) * a.n_patt + h" << x << "] = wtz > 0 ? log(f) + lnscale" << x << " : 0.0; }\n"
              << "               else {\n";
            if (CW > 1) s << "                  if (st == 0) sF[ir * 256 + wv * 64 + " << 16 * g << " + col] = f;\n";
            else s << "                  fh" << x << " += fK[ir] * f;\n";
            s << "                  if (a.want_fhk && own) a.fhK[(cls0 + ir) * a.n_patt + h" << x << "] = wtz > 0 ? f : 0.0; }\n"
              << "            } }\n";
            break;
         default: break;
         }
      }
      if (o.code == OP_MATMUL || o.code == OP_MATMUL_POP) {
         s << "         }\n";
         if (pop >= 0)
            for (int g = 0; g < G; g++) s << "         " << name(out, g) << " = " << name(slot[pop], g) << " * " << name(out, g) << ";\n";
      }
      switch (o.code) {
      case OP_SET_TIP2: if (pl.cherry) ich++; break;
      case OP_PUSH: slot[o.b] = cur; cur = -1; break;
      case OP_MATMUL:
      case OP_MATMUL_POP:
         release(curin);
         if (pop >= 0) { release(slot[pop]); slot[pop] = -1; }
         if (push >= 0) { slot[push] = out; cur = -1; }
         else cur = out;
         break;
      case OP_ROOT: release(cur); cur = -1; break;
      default: break;
      }
   }
   s << "      }\n";      // classes
   if (CW > 1) s << "      __syncthreads();\n";
   for (int g = 0; g < G; g++) {
      const std::string x = gx(g);
      s << "      if (a.mode != PAML_AMD_MODE_LFUN" << (CW > 1 ? " && cw == 0" : "") << ") {\n"
           "         if (a.n_scale) {      /* log-sum-exp around the first maximum (treesub.c:7640-7649) */\n"
           "            const double *fk = a.fhK + cls0 * a.n_patt + (valid" << x << " ? h" << x << " : c_hi - 1);\n"
           "            int it = 0;\n"
           "            for (int ir = 1; ir < K; ir++) if (fk[(long)ir * a.n_patt] > fk[(long)it * a.n_patt]) it = ir;\n"
           "            const double t = fk[(long)it * a.n_patt];\n"
           "            double fh = 0;\n"
           "            for (int ir = 0; ir < K; ir++) fh += fK[ir] * exp(fk[(long)ir * a.n_patt] - t);\n"
           "            v" << x << " = t + log(fh);\n"
           "         }\n"
           "         else {\n";
      if (CW > 1) s << "            for (int ir = 0; ir < K; ir++) fh" << x << " += fK[ir] * sF[ir * 256 + wv * 64 + " << 16 * g << " + col];\n";
      s << "            if (fh" << x << " <= 0) fh" << x << " = 1e-300;\n            v" << x << " = log(fh" << x << ");\n         }\n      }\n";
      s << "      if (st == 0" << (CW > 1 ? " && cw == 0" : "") << ") sV[wv * 64 + " << 16 * g << " + col] = v" << x << ";\n";
   }
   s << "      __syncthreads();\n";
   // the chunk's weighted sum in reduce_stage1's order: thread t adds pattern h0 + t
   s << "      if (" << (CW > 1 ? "cw == 0 && " : "") << "h0 + tid < c_hi) { const double wt = a.weights[h0 + tid]; const double v = wt > 0 ? sV[tid] : 0.0; acc += v * wt; if (a.lnf) a.lnf[(long)bat * a.n_patt + h0 + tid] = v; }\n";
   s << "      __syncthreads();\n";
   s << "   }\n";      // sub-tiles
   if (CW > 1) s << "   if (cw > 0) acc = 0;\n";
   s << "   red_block_finish<" << CW << ">(acc, a.red_partial + (long)bat * a.nb_stride, a.first_chunk + blockIdx.x, a.nb_stride, a.red_out + bat, a.red_counter ? a.red_counter + bat * RED_TICKET_WORDS : nullptr);\n";
   s << "}\n";
   return s.str();
}

// ---- 20 states on v_mfma_f64_4x4x4 (m20_* in device_common.h) -----------------------------------------------------------------
// One class per workgroup (grid = a multiple of the class count, persistent over the 256-pattern tiles of its class): the
// row-major P(t) of every internal branch of that class sits in LDS (3 200 bytes each) for the whole launch; 8 waves x 2 groups
// of 16 patterns; tip rows are gathered from pmat's tables (L1 / L2 resident); fx_r's value per class goes to fhK and the usual
// reduction kernel does the mixture.  One gene; trees whose internal branches fit in LDS (<= 46).
inline bool jit_m20_supported(const Program &p, int n_tips, int n_genes, int *n_slots = nullptr)
{
   (void)n_genes;      // (several genes, round 6: a workgroup serves one (gene, class) — jit_generate_m20)
   if (!jit_valu_supported(p, 64) || n_tips > 64) return false;      // (tip codes: one 16-byte word per lane, 4 lanes per pattern)
   int nmm = 0;
   for (const Op &o : p.ops) {
      if (o.code == OP_MATMUL || o.code == OP_MATMUL_POP) nmm++;
      if (o.code == OP_INIT_TIP) return false;
   }
   if (n_slots) *n_slots = nmm;
   // (trees with more internal branches than LDS holds P(t) blocks for — M20_LDS_NODES — read the others' operands from global memory)
   return nmm >= 1 && nmm <= 200 && p.max_stack + 2 <= 9;
}
constexpr int M20_LDS_NODES = 46;      // P(t) blocks (3 200 bytes each) kept in LDS for the whole launch

// Several genes (G > 1): a gene has its own P(t), so a persistent workgroup serves one (gene, class): the workgroups of a class are dealt
// to the genes in proportion to their 32-pattern units (at least one each), and a gene's workgroups cut ITS units into contiguous ranges.
inline std::string jit_generate_m20(const Program &p, int n_tips, int n_codes, int G = 1)
{
   std::ostringstream s;
   const bool MG = G > 1;
   int nmm = 0;
   std::vector<int> mm_nodes;
   for (const Op &o : p.ops)
      if (o.code == OP_MATMUL || o.code == OP_MATMUL_POP) { mm_nodes.push_back(o.a); nmm++; }
   if (const char *abl = jit_experiment_env("PAML_AMD_M20_ABL")) {      // timing experiments (results are garbage)
      if (strstr(abl, "notip")) s << "#define M20_ABL_NOTIP 1\n";
      if (strstr(abl, "noa")) s << "#define M20_ABL_NOA 1\n";
   }
   s << "#include \"device_common.h\"\nusing namespace paml_amd;\n";
   // experiment (PAML_AMD_M20_W12=1): 12 waves per workgroup, every unit a half unit (one 16-pattern group per wave, three waves per SIMD)
   const bool w12 = getenv("PAML_AMD_M20_W12") != nullptr;
   const int NTH = w12 ? 768 : 512;
   s << "extern \"C\" __global__ __launch_bounds__(" << NTH << ") void prune_jit(PruneArgs a)\n{\n";
   // tip tables: as many as fit beside the P(t) blocks go to LDS (in order of use), the others are gathered from L1 / L2
   const int tip_bytes = n_codes * 168;      // rows padded to 21 doubles in LDS: with 20, codes c and c + 8 share all their banks
   // 16x16x4 + 4x4x4 (m20h_matvec2), else all on 4x4x4 — the latter only for trees whose P(t) all fit in LDS (it reads row-major blocks)
   const bool hybrid = !getenv("PAML_AMD_M20_NOHYBRID") || nmm > M20_LDS_NODES;
   // (experiments: PAML_AMD_M20_NL = P(t) blocks kept in LDS — the rest of LDS takes tip tables; PAML_AMD_M20_GA = products ahead of their
   //  use at which rows gathered from L2 are requested)
   const int lds_nodes = getenv("PAML_AMD_M20_NL") ? std::max(0, std::min(M20_LDS_NODES, atoi(getenv("PAML_AMD_M20_NL")))) : M20_LDS_NODES;
   const int NL = hybrid ? std::min(nmm, lds_nodes) : nmm;      // products 0 .. NL - 1: operands in LDS; the others: in global memory, operand order
   const int room = 158 * 1024 - NL * 3200;
   const int n_lds_max = getenv("PAML_AMD_M20_NOLDSTIP") ? 0 : std::max(0, room / tip_bytes);
   std::vector<int> lds_slot(n_tips, -1);
   int n_lds = 0;
   for (const Op &o : p.ops) {
      auto take = [&](int t) { if (t >= 0 && t < n_tips && lds_slot[t] < 0 && n_lds < n_lds_max) lds_slot[t] = n_lds++; };
      if (o.code == OP_SET_TIP || o.code == OP_MUL_TIP) take(o.a);
      if (o.code == OP_SET_TIP2 || o.code == OP_MUL_TIP2) { take(o.a); take(o.b); }
   }
   s << "   constexpr int NMM = " << std::max(1, NL) << ", NC = " << n_codes << ", NLT = " << std::max(1, n_lds) << ";\n";
   s << "   __shared__ __attribute__((aligned(16))) double sP[NMM * 400];\n";
   s << "   __shared__ __attribute__((aligned(16))) double sT[NLT * NC * 21];\n";
   s << "   __shared__ int sTicket;\n";
   s << "   const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, st = lane >> 4, col = lane & 15;\n";
   if (getenv("PAML_AMD_PROF_TILES")) s << "   if (a.prof && tid == 0) a.prof[(long)blockIdx.x * a.prof_stride + a.prof_stride - 3] = __builtin_amdgcn_s_memrealtime();      /* kernel entry, before the LDS fill */\n";
   if (!MG) s << "   const int iclass = blockIdx.x % a.K, first = blockIdx.x / a.K, stride = gridDim.x / a.K;\n   constexpr int gene = 0, tile0 = 0;\n   const int hbeg = 0;\n   (void)hbeg;\n";
   else {
      // gene g's workgroups (of this class): [g + rest * cum(g) / tot, g + 1 + rest * cum(g + 1) / tot), cum = units of the genes before,
      // rest = workgroups per class beyond one per gene (the host launches at least G per class)
      s << "   const int iclass = blockIdx.x % a.K, wgc = blockIdx.x / a.K, W = gridDim.x / a.K;\n";
      s << "   int gene = 0, first = 0, stride = 1, tile0 = 0;\n";
      s << "   { const CONST_AS int *go = as_const(a.gene_off);\n";
      s << "     long tot = 0; for (int g = 0; g < a.n_genes; g++) tot += (go[g + 1] - go[g] + 31) / 32;\n";
      s << "     if (tot < 1) tot = 1;\n";
      s << "     const long rest = W - a.n_genes; long cu = 0; int t0 = 0;\n";
      s << "     for (int g = 0; g < a.n_genes; g++) {\n";
      s << "        const long ug = (go[g + 1] - go[g] + 31) / 32;\n";
      s << "        const int lo = g + (int)(rest * cu / tot), hi = g + 1 + (int)(rest * (cu + ug) / tot);\n";
      s << "        if (wgc >= lo && wgc < hi) { gene = g; first = wgc - lo; stride = hi - lo; tile0 = t0; }\n";
      s << "        cu += ug; t0 += (go[g + 1] - go[g] + 255) / 256;\n";
      s << "     } }\n";
      s << "   const int hbeg = as_const(a.gene_off)[gene];\n";
      s << "   if (as_const(a.gene_off)[gene + 1] <= hbeg) return;      /* (a shard that holds nothing of this gene) */\n";
   }
   // (a.pint: the branches' P(t) in operand order — [kb][lane] <- P[lane & 15][4 kb + (lane >> 4)], then [kb][k][i] <- P[16 + i][4 kb + k],
   //  written so by pmat_kernel_t<32> in layout 2; a.pcol: the row-major copies, which the all-4x4x4 form reads)
   s << "   const double *Pall = " << (hybrid ? "a.pint" : "a.pcol") << " + ((long)gene * a.K + iclass) * a.n_nodes * 400;\n";
   s << "   const double *Ptip = a.ptip + ((long)gene * a.K + iclass) * a.n_nodes * a.tip_words;\n";
   for (int k = 0; k < NL; k++)
      s << "   for (int i = tid; i < 400; i += " << NTH << ") sP[" << k * 400 << " + i] = Pall[" << (long)mm_nodes[k] * 400 << " + i];\n";
   for (int t = 0; t < n_tips; t++)
      if (lds_slot[t] >= 0)
         s << "   for (int i = tid; i < NC * 20; i += " << NTH << ") sT[" << lds_slot[t] << " * NC * 21 + (i / 20) * 21 + i % 20] = Ptip[(long)" << t << " * a.tip_words + i];\n";
   s << "   __syncthreads();\n";
   s << "   const int aoff = (lane & 3) * 20 + (lane >> 4);      /* A operand: lane 16 k + 4 b + i <- P[4I + i][4K + k] */\n";
   s << "   double pis[5];\n   _Pragma(\"unroll\") for (int m = 0; m < 5; m++) pis[m] = a.pi[(a.n_pi > 1 ? gene : 0) * 20 + 4 * m + st];\n";
   // tip codes: pattern-major, four per dword (PruneArgs::zpm); the NEXT tile's are fetched while this tile is walked, so that no
   // tip row's address waits on a global load
   // (round 6: the four state-quarter lanes of a pattern used to hold the same ZW dwords each — 64 VGPRs for the two groups of a unit and
   //  the next unit's at 60 taxa, and the walk spilled; now lane (st, col) holds the pattern's 16-byte word number st, the codes of tips
   //  16 st .. 16 st + 15, and a tip step fetches its dword from the owner lane with one ds_bpermute_b32: 16 VGPRs whatever the tree)
   const int ZW = ((n_tips + 3) / 4 + 3) / 4 * 4;
   s << "   constexpr int ZW = " << ZW << ";\n";
   s << "   const int hend = as_const(a.gene_off)[gene + 1];\n";
   s << "   const int zq = st < ZW / 4 ? st : ZW / 4 - 1, bp0 = col * 4;\n";
   s << "   unsigned int zn_0[4], zn_1[4];\n";
   // Work inside a workgroup is handed out per wave in units of 32 patterns
   // from an LDS ticket: the two waves of a SIMD do not advance at the same pace (the older wave wins the MFMA arbitration), and
   // with fixed slots the kernel ended 20 % after its fastest waves had finished.  The first unit of a wave is its own slot; the
   // next one is drawn a unit ahead, so that its tip codes arrive while the current unit is walked.
   // Work inside a workgroup is handed out per wave by an LDS ticket, in units of 32 patterns (two 16-pattern groups share every
   // operand fetch): the two waves of a SIMD do not advance at the same pace (the older wave wins the MFMA arbitration, 7 units
   // against 5-6), and with fixed slots the kernel ended 20 % after its fastest waves had finished.  The LAST units of a
   // workgroup's range are handed out as half units — one 16-pattern group, a second copy of the walk without the other group's
   // instructions: when the tickets run out a wave is at most half a unit from its end instead of a whole one
   // (profiles/r03_20state.txt: the tail was 6 % of the span).  Ticket t < nfull: the full unit ubase + t; else the half
   // (t - nfull) & 1 of unit ubase + nfull + (t - nfull) / 2.  A wave's first ticket is its own number; the next one is drawn a
   // unit ahead, so that its tip codes arrive while the current unit is walked.
   // (PAML_AMD_M20_HALF=1, experiment: every unit a half unit — one 16-pattern group per wave, half the partial arrays: deep trees whose
   //  two-group walk spills)
   const int SPLIT = (w12 || getenv("PAML_AMD_M20_HALF")) ? (1 << 20) : (hybrid && !getenv("PAML_AMD_M20_NOSPLIT")) ? 8 : 0;
   s << "#define M20_UNIT_OF(T) ((T) < nfull ? ubase + (T) : ubase + nfull + (((T) - nfull) >> 1))\n";
   s << "#define M20_HALF_OF(T) ((T) < nfull ? -1 : (((T) - nfull) & 1))\n";
   s << "#define M20_FETCH_CODES(T) { int tn_ = (T) < nt ? (T) : nt - 1; tn_ = tn_ < 0 ? 0 : tn_; const int un_ = M20_UNIT_OF(tn_), hf_ = M20_HALF_OF(tn_); \\\n"
        "      const int h0n = as_const(a.tiles)[tile0 + (un_ >> 3)].y + (un_ & 7) * 32 + (hf_ > 0 ? 16 : 0); \\\n"
        "      long hn = h0n + col; if (hn >= hend) hn = hend - 1; const uint4 *zp = (const uint4 *)(a.zpm + hn * ZW); \\\n"
        "      { const uint4 t = zp[zq]; zn_0[0] = t.x; zn_0[1] = t.y; zn_0[2] = t.z; zn_0[3] = t.w; } \\\n"
        "      hn = h0n + 16 + col; if (hn >= hend) hn = hend - 1; zp = (const uint4 *)(a.zpm + hn * ZW); \\\n"
        "      { const uint4 t = zp[zq]; zn_1[0] = t.x; zn_1[1] = t.y; zn_1[2] = t.z; zn_1[3] = t.w; } }\n";
   s << "#define M20_TICKET() __builtin_amdgcn_readfirstlane(lane == 0 ? __hip_atomic_fetch_add(&sTicket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0)\n";
   if (hybrid && NL > 0) s << "   double Acol[5], AsN[5] = {0, 0, 0, 0, 0};\n   (void)AsN;\n   m20h_read_big(m20_lds_addr(sP), lane, Acol);      /* the first product's big operands (a unit's last product fetches them for the next unit) */\n   (void)aoff;\n";
   else if (hybrid)      // (no P(t) block in LDS: the first product's operands come from global memory like every other's)
      s << "   double Acol[5], AsN[5];\n   { const double *Pn_ = Pall + " << (long)mm_nodes[0] * 400 << "; _Pragma(\"unroll\") for (int i = 0; i < 5; i++) { Acol[i] = Pn_[i * 64 + lane]; AsN[i] = Pn_[320 + i * 16 + ((lane >> 4) << 2) + (lane & 3)]; } }\n   (void)aoff; (void)sP;\n";
   else s << "   double Acol[5];\n   m20_acol_asm<0>(m20_lds_addr(sP) + aoff * 8, Acol);      /* first column of the first product (a unit's last product fetches it for the next unit) */\n";
   const bool proft = getenv("PAML_AMD_PROF_TILES") != nullptr;      // experiments: workgroup timeline (tools/prof_tiles.py)
   const char *ptid = getenv("PAML_AMD_PROF_TID");                   // ... stamped by this thread (default 0)
   const std::string pt = ptid ? ptid : "0";
   if (proft) s << "   int ptc = 0; if (a.prof && tid == " << pt << ") { a.prof[(long)blockIdx.x * a.prof_stride] = __builtin_amdgcn_s_memrealtime(); a.prof[(long)blockIdx.x * a.prof_stride + a.prof_stride - 2] = __builtin_amdgcn_s_memtime(); }\n";
   // the class's units (32 patterns each, numbered through its 256-pattern tiles) are cut into one contiguous range per workgroup:
   // 10^5 patterns over 64 workgroups are 48 or 49 units each, where whole tiles were 48 or 56
   s << "   const int total_units = " << (MG ? "(hend - hbeg + 31) / 32" : "min(a.n_tiles * 8, (hend + 31) / 32)") << ";\n";
   s << "   const int ubase = (int)((long)first * total_units / stride), uend = (int)((long)(first + 1) * total_units / stride);\n";
   s << "   const int nfull = max(0, (uend - ubase) - " << SPLIT << "), nt = nfull + 2 * ((uend - ubase) - nfull);\n";
   s << "   if (threadIdx.x == 0) sTicket = " << NTH / 64 << ";      /* the first tickets are the waves' own */\n   __syncthreads();\n";
   s << "   int u = wv, unext = M20_TICKET();\n";
   s << "   M20_FETCH_CODES(u)\n";
   s << "   while (u < nt) {\n";
   s << "      const int unit = M20_UNIT_OF(u), half = M20_HALF_OF(u);\n";
   s << "      const int h0 = as_const(a.tiles)[tile0 + (unit >> 3)].y + (unit & 7) * 32 + (half > 0 ? 16 : 0);\n";
   s << "      unsigned int zw_0[4], zw_1[4];\n";
   s << "      _Pragma(\"unroll\") for (int i = 0; i < 4; i++) { zw_0[i] = zn_0[i]; zw_1[i] = zn_1[i]; }\n";
   s << "      M20_FETCH_CODES(unext)\n";
   s << "      const int unext2 = M20_TICKET();\n";
   auto emit_body = [&](const int G) {      // the walk over one unit: G = 2 pattern groups, or (half units) group 0 alone
   for (int g = 0; g < G; g++) {
      s << "      const long h_" << g << " = h0 + " << 16 * g << " + col;\n      const bool valid_" << g << " = h_" << g << " < hend;\n";
      s << "      double lnscale_" << g << " = 0;\n      (void)lnscale_" << g << ";\n";
   }
   const int NA = p.max_stack + 2;
   for (int g = 0; g < G; g++)
      for (int i = 0; i < NA; i++) s << "      double A" << i << "_" << g << "[5];\n";
   std::vector<int> freeA;
   for (int i = NA - 1; i >= 0; i--) freeA.push_back(i);
   auto alloc = [&]() { int r = freeA.back(); freeA.pop_back(); return r; };
   auto release = [&](int r) { freeA.push_back(r); };
   auto name = [&](int a, int g) { return "A" + std::to_string(a) + "_" + std::to_string(g); };
   const char *LOOP = "_Pragma(\"unroll\") for (int m = 0; m < 5; m++) ";
   // Tip rows are gathered from global memory: the loads of the tip steps that follow a matrix product are issued BEFORE that
   // product (its 50 MFMAs cover their latency) and nothing else is allowed to move across a step (sched_barrier): left to
   // itself the compiler hoists every tip load of the tile to the top and spills.
   const size_t nops = p.ops.size();
   auto is_tip = [&](const Op &o) { return o.code == OP_SET_TIP || o.code == OP_MUL_TIP || o.code == OP_SET_TIP2 || o.code == OP_MUL_TIP2; };
   auto is_mm = [&](const Op &o) { return o.code == OP_MATMUL || o.code == OP_MATMUL_POP; };
   std::vector<char> loaded(nops, 0);
   auto tip_src = [&](int t) {
      return lds_slot[t] >= 0 ? "sT + " + std::to_string(lds_slot[t]) + " * NC * 21, 21" : "Ptip + (long)" + std::to_string(t) + " * a.tip_words, 20";
   };
   auto emit_loads = [&](size_t i) {      // the rows of tip step i -> T<i>a_<g>, T<i>b_<g>
      const Op &o = p.ops[i];
      const bool two = o.code == OP_SET_TIP2 || o.code == OP_MUL_TIP2;
      for (int g = 0; g < G; g++) {
         auto codeof = [&](int t) {
            return "(int)(((unsigned)__builtin_amdgcn_ds_bpermute(bp0 + " + std::to_string((t >> 4) * 64) + ", (int)zw_" + std::to_string(g) + "[" + std::to_string((t >> 2) & 3) + "]) >> " +
                   std::to_string((t & 3) * 8) + ") & 0xffu)";
         };
         s << "      double T" << i << "a_" << g << "[5]; m20_tip(" << tip_src(o.a) << ", " << codeof(o.a) << ", st, T" << i << "a_" << g << ");\n";
         if (two) s << "      double T" << i << "b_" << g << "[5]; m20_tip(" << tip_src(o.b) << ", " << codeof(o.b) << ", st, T" << i << "b_" << g << ");\n";
      }
      loaded[i] = 1;
   };
   // loads run TWO products ahead of their use
   std::vector<size_t> mm_at;
   for (size_t i = 0; i < nops; i++)
      if (is_mm(p.ops[i])) mm_at.push_back(i);
   // ... rows gathered from L1 / L2 run TWO products ahead of their use, rows in LDS one (their latency is a fraction of a
   // product, and every step in flight holds 20 - 40 registers)
   auto all_lds = [&](const Op &o) {
      const bool two = o.code == OP_SET_TIP2 || o.code == OP_MUL_TIP2;
      return lds_slot[o.a] >= 0 && (!two || lds_slot[o.b] >= 0);
   };
   auto emit_loads_after = [&](int k, int which) {      // the tip steps between product k and product k + 1 (k = -1: in front of the first); which: 1 = LDS rows, 2 = the others, 3 = all
      const size_t lo = k < 0 ? 0 : (k < (int)mm_at.size() ? mm_at[k] + 1 : nops), hi = k + 1 < (int)mm_at.size() ? mm_at[k + 1] : nops;
      for (size_t j = lo; j < hi; j++)
         if (is_tip(p.ops[j]) && !loaded[j] && (which & (all_lds(p.ops[j]) ? 1 : 2))) emit_loads(j);
   };
   emit_loads_after(-1, 3);
   const bool ga1 = getenv("PAML_AMD_M20_GA") && atoi(getenv("PAML_AMD_M20_GA")) == 1;
   if (!ga1) emit_loads_after(0, 2);
   std::vector<int> slot(256, -1);
   int cur = -1, imm = 0;
   for (size_t iop = 0; iop < nops; iop++) {
      const Op &o = p.ops[iop];
      int out = -1, pop = -1, push = -1, curin = cur;
      if (o.code == OP_INIT_ONES || o.code == OP_SET_TIP || o.code == OP_SET_TIP2) {
         if (cur < 0) cur = alloc();
         curin = cur;
      }
      switch (o.code) {
      case OP_INIT_ONES:
         for (int g = 0; g < G; g++) s << "      " << LOOP << name(curin, g) << "[m] = 1.0;\n";
         break;
      case OP_SET_TIP:
         for (int g = 0; g < G; g++) s << "      " << LOOP << name(curin, g) << "[m] = T" << iop << "a_" << g << "[m];\n";
         break;
      case OP_MUL_TIP:
         for (int g = 0; g < G; g++) s << "      " << LOOP << name(curin, g) << "[m] *= T" << iop << "a_" << g << "[m];\n";
         break;
      case OP_SET_TIP2:
         for (int g = 0; g < G; g++) s << "      " << LOOP << name(curin, g) << "[m] = T" << iop << "a_" << g << "[m] * T" << iop << "b_" << g << "[m];\n";
         break;
      case OP_MUL_TIP2:
         for (int g = 0; g < G; g++)
            s << "      " << LOOP << name(curin, g) << "[m] = (" << name(curin, g) << "[m] * T" << iop << "a_" << g << "[m]) * T" << iop << "b_" << g << "[m];\n";
         break;
      case OP_PUSH: slot[o.b] = cur; cur = -1; break;
      case OP_MATMUL:
      case OP_MATMUL_POP:
         pop = mm_pop_slot(o); push = mm_push_slot(o); out = alloc();
         emit_loads_after(imm, 1);
         emit_loads_after(ga1 ? imm : imm + 1, 2);
         s << "      __builtin_amdgcn_sched_barrier(0);\n";
         {
            const int nxt = (imm + 1) % nmm;
            auto opnd = [&](int k) { return k < NL ? "sP + " + std::to_string(k * 400) : "Pall + " + std::to_string((long)mm_nodes[k] * 400); };
            const std::string tpl = std::string("<") + (imm >= NL ? "true" : "false") + ", " + (nxt >= NL ? "true" : "false") + ">";
            if (!hybrid && G == 2)
               s << "      m20_matvec2(sP + " << imm * 400 << ", sP + " << nxt * 400 << ", aoff, Acol, " << name(curin, 0) << ", " << name(out, 0) << ", " << name(curin, 1) << ", " << name(out, 1) << ");\n";
            else if (G == 2)
               s << "      m20h_matvec2x" << tpl << "(" << opnd(imm) << ", " << opnd(nxt) << ", lane, Acol, AsN, " << name(curin, 0) << ", " << name(out, 0) << ", "
                 << name(curin, 1) << ", " << name(out, 1) << ");\n";
            else
               s << "      m20h_matvec1x" << tpl << "(" << opnd(imm) << ", " << opnd(nxt) << ", lane, Acol, AsN, " << name(curin, 0) << ", " << name(out, 0) << ");\n";
         }
         s << "      __builtin_amdgcn_sched_barrier(0);\n";
         imm++;
         release(curin);
         if (pop >= 0) {
            for (int g = 0; g < G; g++) s << "      " << LOOP << name(out, g) << "[m] = " << name(slot[pop], g) << "[m] * " << name(out, g) << "[m];\n";
            release(slot[pop]);
            slot[pop] = -1;
         }
         if (push >= 0) { slot[push] = out; cur = -1; }
         else cur = out;
         break;
      case OP_SCALE:
         for (int g = 0; g < G; g++)
            s << "      { const double fac = m20_scale(" << name(curin, g) << "); lnscale_" << g << " += fac;\n"
              << "        if (a.keep && st == 0 && valid_" << g << ") a.scalef[((long)iclass * a.n_scale + " << o.b << ") * a.n_patt + h_" << g << "] = fac; }\n";
         break;
      case OP_ROOT:
         for (int g = 0; g < G; g++)
            s << "      m20_root(a, " << name(curin, g) << ", pis, lnscale_" << g << ", iclass, h_" << g << ", st == 0 && valid_" << g << ");\n";
         release(cur);
         cur = -1;
         break;
      default: break;
      }
   }
   };      // emit_body
   if (w12 || getenv("PAML_AMD_M20_HALF")) emit_body(1);
   else if (SPLIT) {
      s << "      if (half < 0) {\n";
      emit_body(2);
      s << "      } else {\n";
      emit_body(1);
      s << "      }\n";
   }
   else emit_body(2);
   if (proft) s << "      if (a.prof && tid == " << pt << " && ptc < a.prof_stride - 4) { a.prof[(long)blockIdx.x * a.prof_stride + 1 + ptc] = __builtin_amdgcn_s_memrealtime(); a.prof[(long)blockIdx.x * a.prof_stride + a.prof_stride - 1] = __builtin_amdgcn_s_memtime(); }\n      ptc++;\n";
   s << "      u = unext; unext = unext2;\n";
   s << "   }\n}\n";
   return s.str();
}

// ---- small data sets: the cooperative kernel (prune_mfma64_coop) unrolled for one tree, reduction inside (device_common.h, COOPJ_*) ----
// Every operand of the walk is requested straight into registers as far ahead as `budget` VGPRs allow (a product's A operands 32,
// a tip's two row pieces 8): the generator walks the program once, keeping a list of the operands in order of use, and emits a
// request whenever the registers of the requests still open fit the budget — at the top of the kernel until it is full, then
// after every step that consumed one.
inline bool jit_coop_supported(const Program &p, int n_tips, int n_codes)
{
   if (p.ops.size() > 260 || n_tips > 96 || n_codes > 256) return false;
   for (const Op &o : p.ops)
      if (o.code == OP_STORE || o.code == OP_LOAD || o.code == OP_EXPORT) return false;
   return p.max_stack <= 12;
}

inline std::string jit_generate_coop(const Program &p, int n_tips, int n_states = 61, int budget = 380)
{
   std::ostringstream s;
   // k-blocks of four states that hold states of the model, in pairs (a 16-byte word of the operand order): the rest of the padded
   // 64 x 64 matrices is zeros and is neither fetched nor multiplied (20 states: 5 k-blocks instead of 16)
   const int KB = n_states <= 32 ? (n_states + 3) / 4 : 16, NP = (KB + 1) / 2, pcost = 4 * NP;
   s << "#define COOPJ_NP " << NP << "\n#define COOPJ_KB " << KB << "\n";
   s << "#include \"device_common.h\"\nusing namespace paml_amd;\n";
   s << "extern \"C\" __global__ __launch_bounds__(256) void prune_jit(PruneArgs a)\n{\n   COOPJ_PROLOGUE\n";
   s << "   const int cj_bat = iclass / a.Km, cj_nwg = a.n_tiles * 4 * a.Km;\n";
   s << "   if (!empty) {\n";
   // the operands in order of use
   struct Req { bool tip; int id; int cost; };      // id: tip number, or index of the product
   std::vector<Req> reqs;
   std::vector<int> req_of_op_a(p.ops.size(), -1), req_of_op_b(p.ops.size(), -1);
   int n_prod = 0;
   for (size_t i = 0; i < p.ops.size(); i++) {
      const Op &o = p.ops[i];
      switch (o.code) {
      case OP_MATMUL: case OP_MATMUL_POP: req_of_op_a[i] = (int)reqs.size(); reqs.push_back({false, n_prod++, pcost}); break;
      case OP_SET_TIP: case OP_MUL_TIP: req_of_op_a[i] = (int)reqs.size(); reqs.push_back({true, o.a, 8}); break;
      case OP_SET_TIP2: case OP_MUL_TIP2:
         req_of_op_a[i] = (int)reqs.size(); reqs.push_back({true, o.a, 8});
         req_of_op_b[i] = (int)reqs.size(); reqs.push_back({true, o.b, 8});
         break;
      default: break;
      }
   }
   // the character codes of this lane's pattern, every tip (one byte load each, all in flight together)
   std::vector<char> used(n_tips, 0);
   for (const Op &o : p.ops) {
      if (o.code == OP_SET_TIP || o.code == OP_MUL_TIP || o.code == OP_INIT_TIP) used[o.a] = 1;
      if (o.code == OP_SET_TIP2 || o.code == OP_MUL_TIP2) used[o.a] = used[o.b] = 1;
   }
   for (int t = 0; t < n_tips; t++)
      if (used[t]) s << "   const int c" << t << " = (int)a.z[(long)" << t << " * a.z_stride + hc];\n";
   // product k: node of its branch
   std::vector<int> prod_node;
   for (const Op &o : p.ops)
      if (o.code == OP_MATMUL || o.code == OP_MATMUL_POP) prod_node.push_back(o.a);
   size_t next_req = 0;
   int open_cost = 0;
   auto top_up = [&]() {
      bool any = false;
      while (next_req < reqs.size() && (open_cost + reqs[next_req].cost <= budget || open_cost == 0)) {
         const Req &r = reqs[next_req];
         if (r.tip) s << "   COOPJ_T(T" << next_req << ", " << r.id << ", c" << r.id << ")\n";
         else s << "   COOPJ_P(P" << r.id << ", " << prod_node[r.id] << ")\n";
         open_cost += r.cost;
         next_req++;
         any = true;
      }
      if (any) s << "   __builtin_amdgcn_sched_barrier(0);\n";
   };
   auto consumed = [&](int req) {
      // (requests are consumed in the order they were made)
      open_cost -= reqs[req].cost;
   };
   top_up();
   const int NA = p.max_stack + 2;
   for (int i = 0; i < NA; i++) s << "   double A" << i << "[4];\n";
   std::vector<int> freeA;
   for (int i = NA - 1; i >= 0; i--) freeA.push_back(i);
   auto alloc = [&]() { int r = freeA.back(); freeA.pop_back(); return r; };
   auto release = [&](int r) { freeA.push_back(r); };
   auto name = [&](int r) { return "A" + std::to_string(r); };
   std::vector<int> slot(256, -1);
   int cur = -1, xb = 0;
   const char *LOOP = "_Pragma(\"unroll\") for (int r = 0; r < 4; r++) ";
   for (size_t iop = 0; iop < p.ops.size(); iop++) {
      const Op &o = p.ops[iop];
      const int ra = req_of_op_a[iop], rb = req_of_op_b[iop];
      switch (o.code) {
      case OP_INIT_ONES:
         if (cur < 0) cur = alloc();
         s << "   " << LOOP << name(cur) << "[r] = (4 * (4 * wave + r) + q < n) ? 1.0 : 0.0;\n";
         break;
      case OP_INIT_TIP:
         if (cur < 0) cur = alloc();
         s << "   " << LOOP << name(cur) << "[r] = (a.cleandata && 4 * (4 * wave + r) + q == c" << o.a << ") ? 1.0 : 0.0;\n";
         break;
      case OP_SET_TIP:
         if (cur < 0) cur = alloc();
         s << "   " << name(cur) << "[0] = T" << ra << "a.x; " << name(cur) << "[1] = T" << ra << "a.y; " << name(cur) << "[2] = T" << ra << "b.x; " << name(cur) << "[3] = T" << ra << "b.y;\n";
         consumed(ra);
         break;
      case OP_MUL_TIP:
         s << "   " << name(cur) << "[0] *= T" << ra << "a.x; " << name(cur) << "[1] *= T" << ra << "a.y; " << name(cur) << "[2] *= T" << ra << "b.x; " << name(cur) << "[3] *= T" << ra << "b.y;\n";
         consumed(ra);
         break;
      case OP_SET_TIP2:
         if (cur < 0) cur = alloc();
         s << "   " << name(cur) << "[0] = T" << ra << "a.x * T" << rb << "a.x; " << name(cur) << "[1] = T" << ra << "a.y * T" << rb << "a.y; "
           << name(cur) << "[2] = T" << ra << "b.x * T" << rb << "b.x; " << name(cur) << "[3] = T" << ra << "b.y * T" << rb << "b.y;\n";
         consumed(ra); consumed(rb);
         break;
      case OP_MUL_TIP2:
         s << "   " << name(cur) << "[0] = (" << name(cur) << "[0] * T" << ra << "a.x) * T" << rb << "a.x; " << name(cur) << "[1] = (" << name(cur) << "[1] * T" << ra << "a.y) * T" << rb << "a.y; "
           << name(cur) << "[2] = (" << name(cur) << "[2] * T" << ra << "b.x) * T" << rb << "b.x; " << name(cur) << "[3] = (" << name(cur) << "[3] * T" << ra << "b.y) * T" << rb << "b.y;\n";
         consumed(ra); consumed(rb);
         break;
      case OP_PUSH:
         slot[o.b] = cur;
         cur = -1;
         break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         const int pop = mm_pop_slot(o), push = mm_push_slot(o), out = alloc();
         s << "   COOPJ_MATVEC(P" << reqs[ra].id << ", " << name(cur) << ", " << name(out) << ", " << xb << ")\n";
         xb ^= 1;
         consumed(ra);
         release(cur);
         if (pop >= 0) {
            s << "   " << LOOP << name(out) << "[r] = " << name(slot[pop]) << "[r] * " << name(out) << "[r];\n";
            release(slot[pop]);
            slot[pop] = -1;
         }
         if (push >= 0) { slot[push] = out; cur = -1; }
         else cur = out;
      } break;
      case OP_SCALE:
         s << "   COOPJ_SCALE(" << name(cur) << ")\n";
         break;
      case OP_ROOT:
         s << "   COOPJ_ROOT(" << name(cur) << ", " << xb << ")\n";
         xb ^= 1;
         release(cur);
         cur = -1;
         break;
      default: break;
      }
      if (ra >= 0) top_up();
   }
   s << "   }\n   coopj_finish(a, cj_bat, (int)blockIdx.x - cj_bat * cj_nwg, cj_nwg);\n}\n";
   return s.str();
}

inline std::string jit_source_dir()
{
   if (const char *e = getenv("PAML_AMD_CSRC")) return e;      // (a variant library built somewhere else: tools/build_variant.sh, PAML_AMD_LIB)
   Dl_info info;
   if (dladdr((const void *)&jit_source_dir, &info) && info.dli_fname) {
      std::string so = info.dli_fname;                 // .../paml_amd/lib/libpaml_amd.so
      const size_t cut = so.rfind('/');
      const std::string libdir = cut == std::string::npos ? "." : so.substr(0, cut);
      return libdir + "/../csrc";
   }
   return "paml_amd/csrc";
}

// Compile `src` for gfx950 (works without a GPU).  Returns 0 on success; `log` gets the compiler output.
// Code objects are kept on disk, keyed by a hash of the generated source, of the header it includes, of the optimisation
// level and of the hiprtc version: a tree seen before (another run of the same analysis) costs a file read instead of
// seconds of hiprtc (0.2 - 19 s per topology, profiles/r01_big_trees.jsonl).  Two places are looked at:
//   <library dir>/jit/         read-only: code objects built together with the library (__graft_entry__.build() fills it for
//                              the benchmark's trees), so a fresh machine does not start with a compile;
//   the user's cache           read-write: $PAML_AMD_JIT_CACHE, else $XDG_CACHE_HOME/paml_amd/jit, else $HOME/.cache/paml_amd/jit;
//                              PAML_AMD_JIT_CACHE=0 (or empty) switches it off.
inline const char *jit_opt_level() { return getenv("PAML_AMD_JIT_OPT") ? getenv("PAML_AMD_JIT_OPT") : "-O3"; }      // experiments: -O1 / -O2

// Kernels of large trees (the generator marks their source): one basic block of tens of thousands of instructions, on which three
// passes of the compiler are quadratic and gain nothing here — measured on the 192-taxon kernel (340 KB of code), this container's CPU:
// GPU Load and Store Vectorizer 56 s of 82 (every access is already 16 bytes wide), Machine CSE 6 s, Machine Copy Propagation 3 s.
// Without them 23 s, the same registers and 115 spilled dwords against 106 (tools/README.md, profiles/r05_big_trees.txt).
static const char *const JIT_BIG_FLAGS[] = {"-mllvm", "-amdgpu-load-store-vectorizer=0", "-mllvm", "-disable-machine-cse", "-mllvm", "-disable-copyprop"};
inline bool jit_is_big(const std::string &src) { return src.compare(0, 10, "// JIT_BIG") == 0 && !getenv("PAML_AMD_JIT_BIG_DEFAULT_FLAGS"); }
inline std::string jit_strip_big(const std::string &src)      // the same kernel for the compiler's full pipeline (no marker line)
{
   return src.compare(0, 10, "// JIT_BIG") == 0 ? src.substr(src.find('\n') + 1) : src;
}
inline std::vector<std::string> jit_big_flags(const std::string &src)      // (PAML_AMD_JIT_BIG_FLAGS="-mllvm -x ...": experiments)
{
   std::vector<std::string> f;
   if (!jit_is_big(src)) return f;
   if (const char *v = getenv("PAML_AMD_JIT_BIG_FLAGS")) {
      std::istringstream is(v);
      for (std::string w; is >> w;) f.push_back(w);
   }
   else f.assign(std::begin(JIT_BIG_FLAGS), std::end(JIT_BIG_FLAGS));
   return f;
}

inline std::string jit_cache_name(const std::string &src)
{
   unsigned long long h = 1469598103934665603ull;
   auto mix = [&](const std::string &t) { for (unsigned char ch : t) { h ^= ch; h *= 1099511628211ull; } };
   mix(src);
   {  // the header the source includes is part of the program
      FILE *f = fopen((jit_source_dir() + "/device_common.h").c_str(), "rb");
      if (f) { char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof(buf), f)) > 0) mix(std::string(buf, n)); fclose(f); }
   }
   mix(jit_opt_level());
   for (const std::string &o : jit_big_flags(src)) mix(o);
   int major = 0, minor = 0;
   (void)hiprtcVersion(&major, &minor);
   mix("hiprtc" + std::to_string(major) + "." + std::to_string(minor));
   char name[64];
   snprintf(name, sizeof(name), "%016llx.gfx950.hsaco", h);
   return name;
}

inline bool jit_mkdirs(const std::string &dir)      // mkdir -p without a shell
{
   for (size_t i = 1; i <= dir.size(); i++)
      if (i == dir.size() || dir[i] == '/') {
         const std::string sub = dir.substr(0, i);
         if (mkdir(sub.c_str(), 0777) != 0 && errno != EEXIST) return false;
      }
   return true;
}

inline std::string jit_shipped_dir() { return jit_source_dir() + "/../lib/jit"; }

inline std::string jit_user_cache_dir()
{
   const char *c = getenv("PAML_AMD_JIT_CACHE");
   if (c) return (!*c || !strcmp(c, "0")) ? std::string() : std::string(c);
   if (const char *x = getenv("XDG_CACHE_HOME"))
      if (*x) return std::string(x) + "/paml_amd/jit";
   if (const char *hm = getenv("HOME"))
      if (*hm) return std::string(hm) + "/.cache/paml_amd/jit";
   return std::string();
}

inline bool jit_read_file(const std::string &path, std::vector<char> *code)
{
   FILE *f = fopen(path.c_str(), "rb");
   if (!f) return false;
   fseek(f, 0, SEEK_END);
   const long n = ftell(f);
   rewind(f);
   code->resize(n > 0 ? n : 0);
   const bool ok = n > 0 && fread(code->data(), 1, n, f) == (size_t)n;
   fclose(f);
   return ok;
}

inline void jit_write_file(const std::string &dir, const std::string &name, const std::vector<char> &code)
{
   if (dir.empty() || !jit_mkdirs(dir)) return;
   const std::string path = dir + "/" + name, tmp = path + ".tmp" + std::to_string((long)getpid());      // write beside, then rename:
   FILE *f = fopen(tmp.c_str(), "wb");                                                                    // readers never see a partial file
   if (!f) return;
   const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
   fclose(f);
   if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
}

inline int jit_compile_code(const std::string &src, std::vector<char> *code, std::string *log, const char *store_dir = nullptr)
{
   const std::string name = jit_cache_name(src), user = jit_user_cache_dir();
   if (const char *d = getenv("PAML_AMD_JIT_SRC_DIR")) {      // debugging: every source that reaches the compiler (or its cache), by cache name
      if (FILE *f = fopen((std::string(d) + "/" + name + ".hip").c_str(), "wb")) { fwrite(src.data(), 1, src.size(), f); fclose(f); }
   }
   if (!store_dir) {
      if (jit_read_file(jit_shipped_dir() + "/" + name, code)) return 0;
      if (!user.empty() && jit_read_file(user + "/" + name, code)) return 0;
   }
   hiprtcProgram prog;
   if (hiprtcCreateProgram(&prog, src.c_str(), "prune_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
      *log = "hiprtcCreateProgram failed";
      return -1;
   }
   const std::string inc = "-I" + jit_source_dir();
   std::vector<const char *> opts = {"--offload-arch=gfx950", jit_opt_level(), "-std=c++17", inc.c_str()};
   std::vector<std::string> big = jit_big_flags(src);
   for (const std::string &o : big) opts.push_back(o.c_str());
   const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
   size_t ls = 0;
   hiprtcGetProgramLogSize(prog, &ls);
   if (ls > 1) {
      log->resize(ls);
      hiprtcGetProgramLog(prog, &(*log)[0]);
   }
   if (r != HIPRTC_SUCCESS) {
      hiprtcDestroyProgram(&prog);
      return -1;
   }
   size_t cs = 0;
   hiprtcGetCodeSize(prog, &cs);
   code->resize(cs);
   hiprtcGetCode(prog, code->data());
   hiprtcDestroyProgram(&prog);
   jit_write_file(store_dir ? std::string(store_dir) : user, name, *code);
   return 0;
}

// The code object of `src` if it is already on disk (the library's lib/jit or the user's cache): no compilation.
inline bool jit_cached_code(const std::string &src, std::vector<char> *code)
{
   const std::string name = jit_cache_name(src), user = jit_user_cache_dir();
   if (jit_read_file(jit_shipped_dir() + "/" + name, code)) return true;
   return !user.empty() && jit_read_file(user + "/" + name, code);
}

inline int jit_load_code(const std::vector<char> &code, JitKernel *out)
{
   if (hipModuleLoadData(&out->mod, code.data()) != hipSuccess) return -1;
   if (hipModuleGetFunction(&out->fn, out->mod, "prune_jit") != hipSuccess) return -1;
   return 0;
}

// Compile and load.
inline int jit_compile(const std::string &src, JitKernel *out, std::string *log)
{
   std::vector<char> code;
   if (jit_compile_code(src, &code, log) != 0) return -1;
   if (hipModuleLoadData(&out->mod, code.data()) != hipSuccess) {
      *log = "hipModuleLoadData failed";
      return -1;
   }
   if (hipModuleGetFunction(&out->fn, out->mod, "prune_jit") != hipSuccess) {
      *log = "hipModuleGetFunction failed";
      return -1;
   }
   return 0;
}

}  // namespace paml_amd
