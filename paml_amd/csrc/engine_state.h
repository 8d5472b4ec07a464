// engine_state.h — the engine object behind the C ABI (include/paml_amd.h) and the helpers its translation units share:
//   engine_core.hip     create / destroy, the set_* entry points, read-back of P(t) / partials / scale factors, profiling
//   engine_comm.hip     pattern shards over several GPUs: shard bounds, the RCCL communicator, the collective stream
//   engine_eval.hip     one evaluation (batched P(t), fused pruning, reduction) and the entry points built on it
//   engine_branch.hip   branch-local lnL(t), dlnL, ddlnL on resident partials; node posteriors
//   engine_beb.hip      the BEB grid integral
//   engine_jitdbg.hip   per-tree kernel generation without an engine (tests, build-time prebuild)
//   engine_compress.hip site-pattern compression on the device (stand-alone)
// Built for gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is dlopen'ed by paml_amd_comm_* (see Rccl below)

#include "../../include/paml_amd.h"
#include "jit.h"
#include "kernel_args.h"
#include "program.h"

namespace paml_amd {
static_assert(JIT_SCRATCH_BASE == MFMA_RS, "the per-tree kernel addresses the interpreter's overflow-stack scratch");

enum KernelKind { KK_VALU4, KK_VALU5, KK_VALU20, KK_MFMA64 };
constexpr int DMA_WAVES = 8;           // mfma64 "stream" kernel: 128 patterns per workgroup, 1 workgroup per CU
constexpr int GATHER_WAVES = 4;        // mfma64 "gather" kernel: 64 patterns per workgroup, 2 per CU
constexpr int VALU_MAXD_SMALL = 16, VALU_MAXD_20 = 8;

template <typename T>
struct DevBuf {
   T *p = nullptr;
   size_t cap = 0;
   hipError_t ensure(size_t n)
   {
      if (n <= cap) return hipSuccess;
      if (p) (void)hipFree(p);
      p = nullptr;
      cap = 0;
      hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
      if (e == hipSuccess) cap = n;
      return e;
   }
   void release()
   {
      if (p) (void)hipFree(p);
      p = nullptr;
      cap = 0;
   }
};

// Pinned host arena for the small per-evaluation inputs, so the H2D copies are truly asynchronous and
// the source stays valid until an event says the copies are done.
struct Staging {
   char *p = nullptr;
   size_t cap = 0, used = 0;
   hipEvent_t ev = nullptr;
   bool pending = false;
   hipError_t begin(size_t need)
   {
      if (pending) { hipError_t r = hipEventSynchronize(ev); if (r != hipSuccess) return r; pending = false; }
      if (need > cap) {
         if (p) (void)hipHostFree(p);
         p = nullptr; cap = 0;
         hipError_t r = hipHostMalloc((void **)&p, need * 2, hipHostMallocDefault);
         if (r != hipSuccess) return r;
         cap = need * 2;
      }
      if (!ev) { hipError_t r = hipEventCreateWithFlags(&ev, hipEventDisableTiming); if (r != hipSuccess) return r; }
      used = 0;
      return hipSuccess;
   }
   template <typename T>
   T *put(const T *src, size_t n)
   {
      used = (used + 15) & ~(size_t)15;
      T *dst = (T *)(p + used);
      memcpy(dst, src, n * sizeof(T));
      used += n * sizeof(T);
      return dst;
   }
   hipError_t end(hipStream_t s) { pending = true; return hipEventRecord(ev, s); }
   void release()
   {
      if (pending && ev) (void)hipEventSynchronize(ev);
      if (p) (void)hipHostFree(p);
      if (ev) (void)hipEventDestroy(ev);
      p = nullptr; ev = nullptr; cap = 0; pending = false;
   }
};

struct EigenHost {
   int kind = -1, nR = 0;
   double kappa = 0;
   DevBuf<double> U, V, Root, Cijk;
   // warm start of the device decomposition (paml_amd_set_eigen_warm_start): the last decomposition's eigenvectors R^T[64][64], how
   // many decompositions in a row started from a predecessor's, and which states that one left out (pi = 0)
   // Two buffers: a decomposition reads its start from buffers the batch does not write (Rt[rt_cur] of ANY set: the one whose last matrix
   // is nearest, by the signature sig[] of a few weighted row sums) and leaves its own eigenvectors in Rt[rt_cur ^ 1].
   DevBuf<double> Rt[2];
   int rt_cur = 0;
   int warm_run = -1;
   unsigned long long live_mask = 0;
   double sig[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// RCCL, bound at run time: libpaml_amd.so keeps loading on hosts without the collective library (single-GPU use needs none
// of it), and a process that already holds a copy of librccl.so.1 (PyTorch ships one) shares that copy — dlopen matches by
// soname — instead of getting a second one.
struct Rccl {
   void *h = nullptr;
   decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
   decltype(&ncclCommInitRank) CommInitRank = nullptr;
   decltype(&ncclCommDestroy) CommDestroy = nullptr;
   decltype(&ncclAllReduce) AllReduce = nullptr;
   decltype(&ncclGetErrorString) GetErrorString = nullptr;
   std::string err;
   bool load()
   {
      if (h) return true;
      // PAML_AMD_RCCL_LIB names the library instead (an RCCL build elsewhere; the tests' shared-memory stand-in that lets two ranks
      // share one GPU, tests/shim/rccl_shim.cpp)
      const char *names[] = {getenv("PAML_AMD_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
      for (const char *nm : names)
         if (nm && *nm && (h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
      if (!h) { err = std::string("dlopen librccl.so.1: ") + dlerror(); return false; }
      GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
      CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
      CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
      AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
      GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
      if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GetErrorString) {
         err = "librccl.so.1 lacks an expected symbol";
         h = nullptr;
         return false;
      }
      return true;
   }
};
inline Rccl &rccl()
{
   static Rccl r;      // (process-wide on purpose: one binding of the library, no engine state)
   return r;
}

// Patterns per partial sum of the lnL reduction.  A function of the GLOBAL pattern count only: with shards cut at multiples
// of it, every rank's partial sums are entries of one global array whose fixed-order total does not depend on the number
// of ranks (paml_amd_comm_init).  At most ~1024 partials.
inline int red_chunk(long n_global) { return (int)std::max<long>(256, ((n_global + 1023) / 1024 + 255) / 256 * 256); }

// Environment switches, read once when the engine is created.
struct EnvCfg {
   bool force_stream = false;
   bool offload = false;
   bool dual = true;
   bool no_coef_cache = false;      // PAML_AMD_NO_COEF_CACHE (measurements): every eval_branch call forms the coefficients again
   bool no_branch_eig = false;      // PAML_AMD_NO_BRANCH_EIG: eval_branch in the P / dP / ddP form (round 2's kernels) also where the eigen-basis form applies
   bool no_coop = false;            // PAML_AMD_COOP=0: small data sets on the gather kernel (one wave per 16-pattern group) instead of prune_mfma64_coop
   bool no_pipeline = false, force_gather = false, jit_sync = false, jit_strict = false, valu20 = false, no_fused = false, mfma4 = false, tail = false, no_m20 = false;
   int jit_waves = 0, comm_cus = -1, lanes = 0;
   std::string jit_dump, prof_ops;
   int prof_tid = 0;
   bool prof_tiles = false;      // the dump is a workgroup timeline (jit.h proft) instead of per-op stamps
   bool comm_stats = false;
   void read()
   {
      no_pipeline = getenv("PAML_AMD_NO_PIPELINE") != nullptr;
      if (const char *v = getenv("PAML_AMD_COOP")) no_coop = atoi(v) == 0;
      offload = getenv("PAML_AMD_OFFLOAD") != nullptr;      // experiment (measured no faster, profiles/r03_comm_overhead.txt): the reduction of eval_device on the side stream
      if (const char *v = getenv("PAML_AMD_DUAL")) dual = atoi(v) != 0;      // 0: one pruning stream (consecutive evaluations' kernels never overlap)
      force_gather = getenv("PAML_AMD_FORCE_GATHER") != nullptr;
      force_stream = getenv("PAML_AMD_FORCE_STREAM") != nullptr;      // experiments: the stream interpreter also on small data sets
      jit_sync = getenv("PAML_AMD_JIT_SYNC") != nullptr;
      jit_strict = getenv("PAML_AMD_JIT_STRICT") != nullptr;
      valu20 = getenv("PAML_AMD_VALU20") != nullptr;
      no_fused = getenv("PAML_AMD_NO_FUSED") != nullptr;
      mfma4 = getenv("PAML_AMD_MFMA4") != nullptr;
      no_m20 = getenv("PAML_AMD_NO_M20") != nullptr;
      no_branch_eig = getenv("PAML_AMD_NO_BRANCH_EIG") != nullptr;
      no_coef_cache = getenv("PAML_AMD_NO_COEF_CACHE") != nullptr;
      tail = getenv("PAML_AMD_TAIL") != nullptr;
      if (const char *v = getenv("PAML_AMD_COMM_CUS")) comm_cus = atoi(v);
      comm_stats = getenv("PAML_AMD_COMM_STATS") != nullptr;
      if (const char *v = getenv("PAML_AMD_LANES")) lanes = atoi(v);      // evaluations of a run in flight at once (2 .. 4; default 2: three measured 5 % slower, four 25 %)
      if (const char *v = getenv("PAML_AMD_JIT_WAVES")) jit_waves = atoi(v);        // experiment: the last workgroup forms the total instead of a stage-2 launch
      if (const char *v = getenv("PAML_AMD_JIT_DUMP")) jit_dump = v;
      if (const char *v = getenv("PAML_AMD_PROF_OPS")) prof_ops = v;
      if (const char *v = getenv("PAML_AMD_PROF_TID")) prof_tid = atoi(v);
      prof_tiles = getenv("PAML_AMD_PROF_TILES") != nullptr;
   }
};

}  // namespace paml_amd

using namespace paml_amd;

// Worker threads that are still compiling when the PROCESS exits — a caller that never destroys its engine, like the reference with the
// binding patched in, after a run shorter than the compilation: exit() would run the static destructors of hiprtc / comgr under them
// (a segmentation fault after the results were written).  Every job's thread is listed here, and an atexit handler — registered after
// the list's own statics, so it runs before they go — joins what is still running.
inline void worker_threads_list(std::thread *th, bool add)
{
   static std::mutex m;
   static std::vector<std::pair<std::thread *, pid_t>> live;      // (the process that started the thread: a fork()ed child has the list, not the threads)
   static const bool registered = (atexit([]() { worker_threads_list(nullptr, false); }), true);
   (void)registered;
   std::vector<std::thread *> to_join;
   {
      std::lock_guard<std::mutex> lk(m);
      if (th && add) live.emplace_back(th, getpid());
      else if (th) live.erase(std::remove_if(live.begin(), live.end(), [th](const std::pair<std::thread *, pid_t> &e) { return e.first == th; }), live.end());
      else      // the process is exiting
         for (const auto &e : live)
            if (e.second == getpid()) to_join.push_back(e.first);
   }
   for (std::thread *t : to_join)
      if (t->joinable()) t->join();
}

struct paml_amd_engine {
   int n = 0, n_tips = 0, n_patt = 0, max_classes = 0, n_genes = 1;
   unsigned flags = 0;
   KernelKind kk = KK_MFMA64;
   bool mfma_dma = true;     // which mfma64 variant (dma needs n_tips <= MFMA_ZT)
   int mfma_waves = DMA_WAVES;
   int tile_patt = 64;
   hipStream_t stream = nullptr;
   std::string err;
   EnvCfg env;
   int device = 0, n_cu = 0;          // the device the engine was created on, its CU count (persistent grids)
   bool stream_attr_set = false;      // > 64 KB dynamic LDS of prune_mfma64_stream requested on this device
   unsigned long long *d_prof = nullptr;   // PAML_AMD_PROF_OPS stamps
   size_t prof_words = 0;
   int prof_blocks = 0, prof_stride = 0;

   // pattern shards over several GPUs (paml_amd_comm_init): this engine holds patterns [first_patt, first_patt + n_patt) of
   // n_patt_global; the reduction's partial sums live at their global positions and are summed over the ranks
   ncclComm_t comm = nullptr;
   int rank = 0, world = 1;
   long n_patt_global = 0, first_patt = 0;
   int chunk = 256, nb_global = 1, first_chunk = 0;
   DevBuf<double> d_partial_tot, d_btot;
   // The exchange step runs on its own stream `sc`, ordered by events, so that evaluation i + 1 prunes while evaluation i's partial
   // sums are all-reduced and added up: two slots of (partial sums, their all-reduced copy) alternate; slot b's next writer (two
   // evaluations later) waits for ev_done[b].  The caller's stream is joined to the outstanding totals by paml_amd_flush and by
   // every entry point other than paml_amd_eval_device (join_comm).
   hipStream_t sc = nullptr;
   static constexpr int MAXL = 4;     // slots of (class likelihoods, partial sums, all-reduced copy): the evaluations in flight
   int n_lanes = 2;                   // ... in use (PAML_AMD_LANES), taken in rotation
   hipEvent_t ev_part[MAXL] = {}, ev_done[MAXL] = {};
   // Two facts per slot.  done_pending: an exchange step queued on `sc` may still be reading the slot's buffers — whoever writes the
   // slot next makes ITS stream wait for ev_done (wait_slot in launch_eval).  join_pending: the caller's stream has not been made to
   // wait for that total yet (paml_amd_flush / enter).  A flush in the middle of a run clears only the second: the slot's next
   // writer may run on a pruning stream the flush did not touch.
   bool done_pending[MAXL] = {}, join_pending[MAXL] = {};
   // PAML_AMD_COMM_STATS=1 / paml_amd_comm_stats: timed events around the exchange step of the last evaluations of a run (a ring):
   // exchange = partial sums ready -> total formed (all-reduce + stage 2 on `sc`), lane wait = how long the evaluation's pruning
   // stream sat in front of its slot's previous exchange.  Off by default (timed events cost a few microseconds each).
   static constexpr int NSTAT = 64;
   bool comm_stats = false;
   hipEvent_t st_part[NSTAT] = {}, st_done[NSTAT] = {}, st_w0[NSTAT] = {}, st_w1[NSTAT] = {};
   bool st_waited[NSTAT] = {};
   long st_count = 0;
   int red_slot = 0, last_slot = 0, last_fhk = 0;      // last_fhk: the class-likelihood buffer the last evaluation wrote (fhk_slot)
   // CUs the persistent pruning kernels of SINGLE evaluations leave free while the engine has a communicator (runs of evaluations
   // on two pruning streams take every CU, see engine_eval.hip): their workgroups fill a CU (two waves
   // per SIMD at 256 VGPRs, 130 KB of LDS), so the collective's workgroups would otherwise wait for the kernel's tail — or, when
   // they win the race for a CU at its start, hold back one pruning workgroup for as long as the all-reduce waits for its peers.
   // Free at the benchmark's sizes: 10^6 / N patterns in 128-pattern tiles take 31 / 16 / 8 / 4 rounds on 254 CUs as on 256.
   int comm_cus = 2;
   DevBuf<double> d_partial_s[MAXL - 1], d_partial_tot_s[MAXL - 1], d_fhK_s[MAXL - 1];      // slots 1 .. (slot 0: d_partial, d_partial_tot, d_fhK)
   DevBuf<double> &part_slot(int b) { return b ? d_partial_s[b - 1] : d_partial; }
   DevBuf<double> &tot_slot(int b) { return b ? d_partial_tot_s[b - 1] : d_partial_tot; }
   DevBuf<double> &fhk_slot(int b) { return b ? d_fhK_s[b - 1] : d_fhK; }
   DevBuf<unsigned int> d_zpm;        // fused 4 / 5-state kernel: tip codes pattern-major
   int zpm_words = 0;
   DevBuf<int> d_red_counter;         // "last workgroup adds up the partial sums" tickets, one per batch element
   long bpart_rows = 0; int bpart_cols = 0;      // shape of the last eval_branch's partial-sum array
   bool bpart_colmajor = false;                  // ... stored [column][row] (the eigen-basis kernels), handed out [row][column] either way
   double *h_out = nullptr;           // pinned, device-visible: the synchronous entry points have lnL written straight to the host
   size_t h_out_cap = 0;
   bool fused = false;                // the selected kernel forms the reduction itself
   bool fused_mfma4 = false;
   bool rate_per_gene = false;      // paml_amd_set_gene_class_rates: class rates [n_genes][K]
   std::vector<double> class_rate;  // the [K] rates of set_classes (what set_gene_class_rates(NULL) goes back to)
   bool want_m20 = false, m20 = false;      // 20 states on v_mfma_f64_4x4x4 (jit_generate_m20)
   int fused_threads = 256;
   bool pmat_valid = false;           // d_rowmajor holds the P(t) of an evaluation in the tree's own orientation
   // branch-local evaluation: resident partials on both sides of every edge, re-used from call to call (eval_branch)
   struct BranchCache {
      bool valid = false;
      int K = 0;
      std::vector<int> up;            // up[v]: the neighbour v's stored partial looks away from
      std::vector<char> ok;           // the stored partial of internal node v is current
      std::vector<double> br, gr;     // branch lengths (by lower node) and gene rates the partials were formed with
      // eigen-basis form (kernels_branch.h): the coefficients c_k of the branch `coef_node` are in d_bl_coef, formed from the current
      // partials of its two ends — further trial lengths on that branch need no matrix product
      bool coef_ok = false;
      int coef_node = -1;
      std::vector<char> frag_ok;      // per branch label: V / U^T diag(pi) in operand order and the tips' z rows are in d_bl_efrag / d_bl_ztab
   } bl;
   DevBuf<double> d_bl_partials, d_bl_scalef, d_bl_frag, d_bl_coef, d_bl_efrag, d_bl_ztab, d_bl_etab, d_bl_ecol;
   bool beig_attr_set = false;
   bool bl_gr_sent = false;           // d_gene_rate holds the gene rates of the branch cache (cleared by whoever else writes the buffer)
   hipEvent_t ev_bk[2] = {};          // paml_amd_profile on: around the contraction kernel(s) of the last eval_branch (paml_amd_branch_kernel_ms)
   bool bk_timed = false;
   long n_branch_coef_hits = 0;      // eval_branch calls served from the stored coefficients
   long n_branch_refill_jit = 0;     // eval_branch calls whose forest of dirty subtrees ran on a per-tree kernel (a refill, engine_branch.hip)
   DevBuf<unsigned long long> d_code_mask;      // per character code: bit s = state s belongs to it
   long n_branch_eval = 0, n_branch_nodes = 0;

   // data
   bool have_tips = false, have_tree = false, have_pi = false, have_classes = false;
   int cleandata = 1, n_codes = 0;
   DevBuf<unsigned char> d_z, d_chara_map, d_is_leaf, d_ztiles;
   int zt_bytes = 0;
   DevBuf<int> d_n_chara, d_gene_off, d_label, d_eigen_of, d_b_eigen_of;
   DevBuf<int2> d_tiles, d_tiles_full;   // tile table of the selected kernel / of the full (gather or valu) kernel
   int n_tiles_full = 0;
   DevBuf<int> d_ztip_of;               // half mode of the per-tree kernel's code blocks (jit_zplan): row -> tip
   std::string zt_key;                  // ... of the program the rows were laid out for ("" = tip order)
   DevBuf<int> d_tile_group0;            // mfma64: per tile of the selected kernel, the resident-partial group of its first 16 patterns (PruneArgs::tile_group0)
   // The tables of the OTHER tile size once built: a keep-partials engine alternates between the per-tree kernel (128-pattern tiles: full
   // evaluations) and the interpreter (64: eval_dirty's LOAD programs it has no kernel for yet, node posteriors) — the change-over swaps
   // the two sets instead of uploading, re-laying the code blocks and synchronising twice (build_tiles).  Dropped by set_tips.
   struct TileStash {
      int built_for = 0;                 // the tile size the set was built for (0: nothing there)
      int n_tiles = 0, zt_bytes = 0;
      std::string zt_key;
      DevBuf<int2> tiles;
      DevBuf<int> group0, ztip_of;
      DevBuf<unsigned char> ztiles;
   } tile_stash;
   int tiles_built_for = 0;              // the tile size the current set (d_tiles, d_tile_group0, d_ztiles ...) was built for
   void swap_tile_stash()
   {
      TileStash &t = tile_stash;
      std::swap(t.built_for, tiles_built_for); std::swap(t.n_tiles, n_tiles); std::swap(t.zt_bytes, zt_bytes); std::swap(t.zt_key, zt_key);
      std::swap(t.tiles, d_tiles); std::swap(t.group0, d_tile_group0); std::swap(t.ztip_of, d_ztip_of); std::swap(t.ztiles, d_ztiles);
   }
   int part_groups() const { return n_tiles_full * GATHER_WAVES; }      // 16-pattern groups per (class, node) of the resident partials: the 64-pattern tile table's
   DevBuf<double> d_pi_plain;
   DevBuf<double> d_weights, d_pi, d_freqK, d_rate, d_qfactor, d_branch, d_gene_rate;
   std::vector<int> gene_off;
   int n_tiles = 0, n_pi = 1;

   TreeDesc tree;
   Program prog;
   bool prog_valid = false;
   DevBuf<Op> d_ops;
   DevBuf<int> d_stream;
   Staging stage;
   JitKernel jit;            // per-tree specialised kernel (jit.h), valid when jit.fn != nullptr
   // Kernels of this engine's OTHER programs, kept loaded (round 6): a keep-partials engine alternates between its full program and the LOAD
   // programs of paml_amd_eval_dirty (com.oldconP: one per set of clean nodes, treespace.c:250), eval_branch between the tree seen from a
   // branch and the ordinary one — the kernel being left is retired here instead of unloaded, and recalled by its key.
   std::vector<JitKernel> jit_pool;
   std::vector<std::pair<std::string, int>> jit_seen;      // LOAD programs: how often a key has been asked for (a kernel is compiled from the second time on)
   void jit_retire()
   {
      if (jit.mod && jit.fn) {
         if (jit_pool.size() >= 8) { (void)hipModuleUnload(jit_pool.front().mod); jit_pool.erase(jit_pool.begin()); }
         jit_pool.push_back(jit);
      }
      else if (jit.mod) (void)hipModuleUnload(jit.mod);
      jit = JitKernel();
   }
   bool jit_recall(const std::string &key)
   {
      if (jit.fn && jit.key == key) return true;
      for (size_t i = 0; i < jit_pool.size(); i++)
         if (jit_pool[i].key == key && jit_pool[i].fn) {
            const JitKernel k = jit_pool[i];
            jit_pool.erase(jit_pool.begin() + i);
            jit_retire();
            jit = k;
            return true;
         }
      return false;
   }
   int jit_count_request(const std::string &key)
   {
      for (auto &kv : jit_seen)
         if (kv.first == key) return ++kv.second;
      if (jit_seen.size() >= 64) jit_seen.erase(jit_seen.begin());
      jit_seen.emplace_back(key, 1);
      return 1;
   }
   bool jit_enabled = false, use_jit = false;
   bool small20 = false;     // 20 states on the MFMA interpreters because the data set is small (engine_core.hip): not as a shard of a larger one
   bool coop = false;        // the last evaluation ran prune_mfma64_coop (small data sets: four waves per 16-pattern group)
   // ... or its per-tree form with the reduction inside (jit.h: jit_generate_coop).  A module of its own: an engine goes back and forth
   // between it (an evaluation, a small batch) and the interpreters (a batched gradient too large for one 16-pattern group per CU).
   // The kernel is compiled on a worker thread unless its code object is already on disk (lib/jit or the user's cache) or the caller
   // asked to wait (PAML_AMD_JIT flag / PAML_AMD_JIT_SYNC); the interpreter form serves until it is there.  PAML_AMD_COOPJIT=0 / PAML_AMD_JIT=0: never.
   JitKernel jit_coop;
   bool coopj = false, coopj_enabled = true;
   // Consecutive paml_amd_eval_device calls (the loop of a benchmark or of an optimiser's independent evaluations) build the
   // NEXT evaluation's P(t) on a side stream while the previous pruning kernel is still running: its few workgroups fit the CUs
   // that go idle in that kernel's last round.  Two sets of P buffers alternate; the side stream waits for everything the main
   // stream had queued before the previous evaluation (the last readers of the set it is about to overwrite).  Any other API
   // call switches the fast path off until the next eval_device has run in order.
   bool pipe_ok = false;
   hipStream_t s2 = nullptr;
   hipEvent_t ev_entry[2] = {nullptr, nullptr}, ev_pmat = nullptr;
   int entry_sel = 0;
   bool have_prev_entry = false;
   // P(t) storage of the runs of eval_device calls: NPSET sets in rotation (the current one in d_rowmajor / d_pint / d_ptip / d_pcol,
   // the others in `spare`, oldest first from spare_head), so that the side stream can build the P(t) of the evaluations to come
   // while several pruning kernels are in flight (pruning streams, below) — it only has the CUs those leave it.
   static constexpr int NPSET = 6;
   struct PSet { DevBuf<double> rowmajor, pint, ptip, pcol; int id = 0; } spare[NPSET - 1];
   int spare_head = 0;
   DevBuf<double> d2_branch, d2_gene_rate;
   // The persistent pruning kernels (per-tree MFMA kernel, 20-state matrix-core kernel) hold every CU until their last round of
   // tiles, which is rarely full, and a kernel boundary + the two reduction kernels + their stream events (~20-25 us) sit between
   // two of them on one stream.  From the second eval_device of a run on, the evaluations therefore ALTERNATE between the engine's
   // stream and a second pruning stream `sb` (lane = reduction slot: its own class likelihoods and partial sums, the total on
   // `sc`): the workgroups of evaluation i + 1 are queued while evaluation i runs and take each CU the moment it is released.
   // Same kernels, same arguments, same bits; PAML_AMD_DUAL=0 goes back to one stream.  ev_setread[s]: the last pruning kernel
   // that read P set s so far (all the side stream waits for before it overwrites the set: P(t) runs up to two evaluations ahead).
   hipStream_t sb[MAXL - 1] = {};      // pruning streams of lanes 1 .. (lane 0: the engine's stream)
   hipEvent_t ev_setread[NPSET] = {};
   bool setread_rec[NPSET] = {};
   int pset = 0;                 // id of the current P set (spare[].id: of the others)
   bool dual_ok = false, dual_run = false;      // dual_run: the run's first two-stream evaluation (which still waits for the run's uploads) is behind us
   bool jit_forced = false;  // asked for by flag / environment (as opposed to switched on by the problem's size)
   // a large tree's kernel takes many seconds to compile: that happens on a worker thread while the interpreter kernels
   // serve the evaluations, and the engine changes over when the code object is there
   struct JitJob {
      std::thread th;
      std::atomic<int> state{0};      // 0 idle, 1 compiling, 2 code ready, 3 failed
      int stage = 2;                  // large trees: 1 = the quick build (JIT_BIG_FLAGS), 2 = the full one that replaces it
      std::string key, src, log;
      std::vector<char> code;
      JitJob() { worker_threads_list(&th, true); }
      ~JitJob() { if (th.joinable()) th.join(); worker_threads_list(&th, false); }
   };
   std::unique_ptr<JitJob> jit_job, coop_job, bjit_job;      // (bjit: the branch-local evaluation's refill program, engine_branch.hip)
   std::string bjit_failed_key;
   std::string jit_failed_key, coop_failed_key;
   int jit_stage = 0;                 // which build of the large tree's kernel `jit` holds (0: none / a kernel compiled while the caller waited)
   std::string jit_stage2_failed_key;

   std::vector<EigenHost> eigen;
   std::vector<int> h_eigen_of;      // the class table's eigen set ids as last set (set_classes): checked against the sets that exist
   std::vector<double> h_qfactor;    // ... and its Qfactors [K][n_labels]
   DevBuf<PmatRes> d_pres;           // PmatArgs::res: the resolved (parameter set, node) table of single evaluations
   bool pres_valid = false;          // ... is current (dropped by set_tree / set_classes / any set_eigen_*)
   int plain_codes = 0;             // set_tips: codes below this are single states equal to the code
   bool amb_ascending = true;       // set_tips: every code from 64 on lists its states in ascending order (what the per-tree kernel's overflow path sums in)
   int pmat_B = 1;                   // batch elements of the evaluation whose P(t) the buffers hold (get_pmat: element 0)
   bool rowmajor_valid = false;      // d_rowmajor holds the last evaluation's matrices (pmat_mfma_kernel in the mfma64 layout does not write them)
   DevBuf<EigenDev> d_eigen;
   // batched decomposition on the device (paml_amd_set_eigen_qrev_batch): inputs and the table of the sets' buffer pointers
   DevBuf<double> d_eq_q, d_eq_pi, d_eq_scale;
   DevBuf<double *> d_eq_ptr;
   DevBuf<int> d_eq_rc;      // (row, col) of the elements of a sparse hand-over
   std::vector<int> h_eq_rc; // ... as the device holds them
   DevBuf<int> d_eq_sweeps;
   int *h_eig_fail = nullptr;      // pinned, device-visible: a decomposition that hit its sweep limit sets it (eigen_fail_check)
   bool eigen_attr_set = false;
   bool eigen_warm = false;      // paml_amd_set_eigen_warm_start
   long n_eigen_warm = 0;
   long n_eigen_device = 0;
   int eq_last_batch = 0;
   bool eigen_dirty = true;

   int mode = PAML_AMD_MODE_LFUN, K = 1, n_labels = 1;

   // per-evaluation buffers
   DevBuf<double> d_b_qfactor, d_b_freqK, d_b_rate;
   DevBuf<double> d_beb_f, d_beb_part, d_beb_g, d_beb_out, d_beb_pcl;      // BEB grid integral
   DevBuf<double> d_adg_all;      // eval_adg with pattern shards: [K + 1][n_patt_global] gathered class likelihoods and weights
   DevBuf<int> d_beb_iw;
   DevBuf<double> d_rowmajor, d_pint, d_ptip, d_pcol, d_fhK, d_fscale, d_lnf, d_partial, d_out, d_partials, d_scalef, d_stack;
   DevBuf<double> d_expA, d_expB, d_expSA, d_expSB, d_deriv, d_tt, d_bpartial, d_bout;   // branch-local evaluation
   DevBuf<int> d_label_eff;
   DevBuf<Op> d_ops_tmp;
   bool partials_valid = false;

   // profiling
   bool profiling = false;
   std::vector<hipEvent_t> ev_pool;
   std::vector<hipEvent_t> ev_used;   // sextuples per eval
   long prof_evals = 0;
   long n_eval = 0, n_pmat = 0;

   ~paml_amd_engine()
   {
      for (auto &e : eigen) { e.U.release(); e.V.release(); e.Root.release(); e.Cijk.release(); }
      stage.release();
      if (sc) (void)hipStreamSynchronize(sc);      // (no collective may still be in flight when its communicator goes)
      if (comm && rccl().CommDestroy) (void)rccl().CommDestroy(comm);
      if (sc) (void)hipStreamDestroy(sc);
      for (int b = 0; b < MAXL; b++) {
         if (ev_part[b]) (void)hipEventDestroy(ev_part[b]);
         if (ev_done[b]) (void)hipEventDestroy(ev_done[b]);
      }
      for (hipEvent_t ev : ev_bk)
         if (ev) (void)hipEventDestroy(ev);
      for (int i = 0; i < NSTAT; i++)
         for (hipEvent_t ev : {st_part[i], st_done[i], st_w0[i], st_w1[i]})
            if (ev) (void)hipEventDestroy(ev);
      if (h_out) (void)hipHostFree(h_out);
      if (h_eig_fail) (void)hipHostFree(h_eig_fail);
      for (int ln = 0; ln < MAXL && d_prof && env.prof_tiles && prof_words; ln++) {      // the last launch's workgroup timeline (of each pruning stream: <dump>, <dump>.1 ..)
         std::vector<unsigned long long> hp(prof_words);
         if (hipMemcpy(hp.data(), d_prof + (size_t)ln * prof_words, prof_words * 8, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *f = fopen((env.prof_ops + (ln ? "." + std::to_string(ln) : std::string())).c_str(), "wb")) {
               const int hdr[2] = {prof_blocks, prof_stride};
               fwrite(hdr, sizeof(int), 2, f);
               std::vector<int> codes(prof_stride - 3, 0);
               fwrite(codes.data(), sizeof(int), codes.size(), f);
               fwrite(hp.data(), 8, hp.size(), f);
               fclose(f);
            }
      }
      if (d_prof) (void)hipFree(d_prof);
      if (jit.mod) (void)hipModuleUnload(jit.mod);
      for (JitKernel &k : jit_pool)
         if (k.mod) (void)hipModuleUnload(k.mod);
      if (jit_coop.mod) (void)hipModuleUnload(jit_coop.mod);
      for (auto ev : ev_pool) (void)hipEventDestroy(ev);
      for (auto ev : ev_used) (void)hipEventDestroy(ev);
      DevBuf<unsigned char> *b1[] = {&d_z, &d_chara_map, &d_is_leaf, &d_ztiles};
      for (auto b : b1) b->release();
      DevBuf<int> *b2[] = {&d_n_chara, &d_gene_off, &d_label, &d_eigen_of, &d_b_eigen_of, &d_beb_iw};
      for (auto b : b2) b->release();
      d_tiles.release();
      d_tiles_full.release();
      d_tile_group0.release();
      tile_stash.tiles.release(); tile_stash.group0.release(); tile_stash.ztip_of.release(); tile_stash.ztiles.release();
      d_ztip_of.release();
      d_zpm.release();
      d_red_counter.release();
      d_bl_partials.release(); d_bl_scalef.release(); d_bl_frag.release(); d_code_mask.release();
      d_bl_coef.release(); d_bl_efrag.release(); d_bl_ztab.release(); d_bl_etab.release(); d_bl_ecol.release();
      d_ops.release();
      d_ops_tmp.release();
      d_label_eff.release();
      d_stream.release();
      d_eigen.release();
      d_pres.release();
      d_eq_q.release(); d_eq_pi.release(); d_eq_scale.release(); d_eq_ptr.release(); d_eq_rc.release(); d_eq_sweeps.release();
      d_pi_plain.release();
      DevBuf<double> *b3[] = {&d_weights, &d_pi, &d_freqK, &d_rate, &d_qfactor, &d_branch, &d_gene_rate, &d_rowmajor,
                              &d_pint, &d_ptip, &d_pcol, &d_fhK, &d_fscale, &d_lnf, &d_b_qfactor, &d_b_freqK, &d_b_rate, &d_beb_f, &d_beb_part, &d_beb_g, &d_beb_out, &d_beb_pcl, &d_adg_all, &d_partial, &d_out, &d_partials, &d_scalef, &d_stack, &d2_branch, &d2_gene_rate, &d_partial_tot, &d_btot,
                              &d_expA, &d_expB, &d_expSA, &d_expSB, &d_deriv, &d_tt, &d_bpartial, &d_bout};
      for (auto b : b3) b->release();
      for (auto &sp : spare) { sp.rowmajor.release(); sp.pint.release(); sp.ptip.release(); sp.pcol.release(); }
      for (int b = 0; b < MAXL - 1; b++) { d_partial_s[b].release(); d_partial_tot_s[b].release(); d_fhK_s[b].release(); }
   }
};

namespace paml_amd {

inline int fail(paml_amd_engine *e, int code, const std::string &msg)
{
   if (e) e->err = msg;
   return code;
}

#define HIPCHK(call)                                                                                         \
   do {                                                                                                      \
      hipError_t _r = (call);                                                                                \
      if (_r != hipSuccess)                                                                                  \
         return fail(e, _r == hipErrorOutOfMemory ? PAML_AMD_ENOMEM : PAML_AMD_EHIP,                         \
                     std::string(#call) + ": " + hipGetErrorString(_r));                                     \
   } while (0)

template <typename T>
inline hipError_t upload(DevBuf<T> &b, const T *src, size_t n, hipStream_t s)
{
   hipError_t r = b.ensure(n);
   if (r != hipSuccess) return r;
   if (n == 0) return hipSuccess;
   // pageable source: the runtime stages the copy, so the host buffer may be reused on return
   return hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, s);
}

inline int tipw(const paml_amd_engine *e) { return e->kk == KK_MFMA64 ? 64 : e->n; }
// doubles per tip table: mfma64 pads tables of <= 64 codes to one 32 KB stream block
inline size_t tip_words(const paml_amd_engine *e)
{
   if (e->kk != KK_MFMA64) return (size_t)e->n_codes * e->n;
   return e->n_codes <= 64 ? 4096 : (size_t)e->n_codes * 64;
}
inline int pint_words(const paml_amd_engine *e) { return e->kk == KK_MFMA64 ? 4096 : e->n * e->n; }

inline hipEvent_t get_event(paml_amd_engine *e)
{
   hipEvent_t ev;
   if (!e->ev_pool.empty()) {
      ev = e->ev_pool.back();
      e->ev_pool.pop_back();
   }
   else if (hipEventCreate(&ev) != hipSuccess)
      return nullptr;
   e->ev_used.push_back(ev);
   return ev;
}

inline void mark_on(paml_amd_engine *e, hipStream_t s)
{
   if (!e->profiling) return;
   hipEvent_t ev = get_event(e);
   if (ev) (void)hipEventRecord(ev, s);
}
inline void mark(paml_amd_engine *e) { mark_on(e, e->stream); }

// The engine's own streams are created at the HIGHEST priority: their kernels are the small ones (P(t), partial sums, the exchange
// step) that should take a CU the moment one is released, ahead of the next evaluation's queued pruning workgroups; and the runtime
// keeps a pool of hardware queues per priority level, so they do not share a queue with the caller's (normal-priority) stream.
// Measured on MI355X: the same per-evaluation times as with normal priority (profiles/r03_dual_stream.txt); streams with a CU mask
// (a hardware queue of their own each) are 3x slower; a LOWEST-priority side stream is not served while the main stream has work
// queued (profiles/r03_comm_overhead.txt).
inline hipError_t create_engine_stream(hipStream_t *s)
{
   int least = 0, greatest = 0;
   if (getenv("PAML_AMD_STREAM_PRIO_NORMAL") || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess)      // (experiments)
      return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
   if (hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest) == hipSuccess) return hipSuccess;
   (void)hipGetLastError();
   return hipStreamCreateWithFlags(s, hipStreamNonBlocking);      // (a runtime without stream priorities: the measured times are the same)
}

// The engine's side stream (reductions of consecutive eval_device calls, the exchange step over the ranks) and its events.
inline int ensure_side_stream(paml_amd_engine *e)
{
   if (e->sc) return 0;
   hipStream_t sc = nullptr;
   if (create_engine_stream(&sc) != hipSuccess) return fail(e, PAML_AMD_EHIP, "hipStreamCreate(side stream)");
   for (int b = 0; b < paml_amd_engine::MAXL; b++)
      if ((!e->ev_part[b] && hipEventCreateWithFlags(&e->ev_part[b], hipEventDisableTiming) != hipSuccess) ||
          (!e->ev_done[b] && hipEventCreateWithFlags(&e->ev_done[b], hipEventDisableTiming) != hipSuccess)) {
         (void)hipStreamDestroy(sc);      // (a later call starts over: `sc` is only published with all its events)
         return fail(e, PAML_AMD_EHIP, "hipEventCreate(side stream)");
      }
   e->sc = sc;
   return 0;
}

// The caller's stream waits for the totals still on their way on the side stream (no-op when there are none).
inline int join_comm(paml_amd_engine *e)
{
   for (int b = 0; b < paml_amd_engine::MAXL; b++)
      if (e->join_pending[b]) {
         e->join_pending[b] = false;      // (done_pending stays: the slot's next writer may be another stream, see launch_eval's wait_slot)
         if (hipStreamWaitEvent(e->stream, e->ev_done[b], 0) != hipSuccess) return fail(e, PAML_AMD_EHIP, "hipStreamWaitEvent(collective stream)");
      }
   return 0;
}
// Every entry point except paml_amd_eval_device starts here: the fast path of consecutive eval_device calls ends, the stream is
// joined to the collective stream.
inline void enter(paml_amd_engine *e)
{
   if (!e) return;
   e->pipe_ok = false;
   (void)join_comm(e);
}

// (The host's wait at the end of a synchronous entry point is a plain hipStreamSynchronize: polling hipStreamQuery instead was
//  measured in round 4 and is no faster — 81 - 83 us against 78 - 79 per evaluation of 13 taxa x 79 codon patterns: the runtime's own
//  wait already spins.)
// Pinned host memory the kernels can write: the synchronous entry points get their scalars without a device-to-host copy.
inline int ensure_hout(paml_amd_engine *e, size_t n)
{
   if (n <= e->h_out_cap) return 0;
   if (e->h_out) (void)hipHostFree(e->h_out);
   e->h_out = nullptr; e->h_out_cap = 0;
   HIPCHK(hipHostMalloc((void **)&e->h_out, std::max<size_t>(n, 64) * sizeof(double), hipHostMallocDefault));
   e->h_out_cap = std::max<size_t>(n, 64);
   return 0;
}

// After the host synchronisation of an evaluation: did a device eigen-decomposition queued in front of it reach its sweep limit?
// The likelihood just formed then came from unconverged eigenvectors: an error instead of a number.
inline int eigen_fail_check(paml_amd_engine *e)
{
   if (!e->h_eig_fail || !*(volatile int *)e->h_eig_fail) return 0;
   *(volatile int *)e->h_eig_fail = 0;
   return fail(e, PAML_AMD_ENOCONV, "a device eigen-decomposition (set_eigen_qrev_batch) reached its sweep limit without converging: "
                                    "decompose on the host and pass U, V, Root with paml_amd_set_eigen_uvroot (paml_amd_eigen_counters names the sets of the last batch)");
}

}  // namespace paml_amd

// ---- defined in engine_eval.hip, the translation unit that holds the P(t), pruning and reduction kernels -------------------
namespace paml_amd {

// Batched evaluations: B parameter sets run as K*B classes of one launch (class index = b*K + iclass), so the pruning
// kernels are unchanged; P(t) and the reduction index the per-element inputs.  Null members = shared set_classes values.
struct BatchSpec {
   int B;
   const int *eigen_of;      // [B][n_genes][K][n_labels]
   const double *qfactor;    // [B][K][n_labels]
   const double *freqK;      // [B][K]
   const double *rate;       // [B][K]
};

int build_tiles(paml_amd_engine *e);
int select_tiles(paml_amd_engine *e, bool big_tiles, int want_waves, bool jit_ok);
int launch_eval(paml_amd_engine *e, const double *branch, const double *gene_rate, const unsigned char *clean, double *d_lnL_out, bool want_lnf,
                const BatchSpec *bs = nullptr, bool want_pipe = false, bool want_fhk = true);
// launches for the other translation units (a __global__ function has one home)
void launch_pmat(const PmatArgs &pa, const InlineVec &iv, int n_nodes, int psets, bool small, hipStream_t s, bool mfma = false);
bool pmat_on_matrix_cores(const paml_amd_engine *e, const PmatArgs &pa);
void launch_prune_full(paml_amd_engine *e, int max_stack, int n_blocks, const PruneArgs &pr, hipStream_t s);      // gather (21..64 states) or valu interpreter
void launch_zpm(const unsigned char *z, long z_stride, int n_tips, int n_patt, int zw, unsigned int *out, hipStream_t s);

// the table of eigen systems as the kernels read it (uploaded when a set_eigen_* call has changed one)
// (the table may have holes — set ids are the caller's, e.g. one range per parameter vector of a batched gradient with only the changed
//  systems set: what must exist is every set the class table REFERS to, eigen_refs_ok)
inline int eigen_table(paml_amd_engine *e, std::vector<EigenDev> &tab)
{
   tab.resize(e->eigen.size());
   for (size_t i = 0; i < e->eigen.size(); i++) {
      const EigenHost &h = e->eigen[i];
      tab[i] = EigenDev{h.kind, h.nR, h.kappa, h.U.p, h.V.p, h.Root.p, h.Cijk.p};
   }
   return 0;
}
inline int eigen_refs_ok(paml_amd_engine *e, const int *ids, size_t cnt, const char *who)
{
   for (size_t i = 0; i < cnt; i++) {
      if (ids[i] < 0 || ids[i] >= (int)e->eigen.size()) return fail(e, PAML_AMD_EINVAL, std::string(who) + ": eigen_of entry out of range");
      if (e->eigen[ids[i]].kind < 0) return fail(e, PAML_AMD_EINVAL, "eigen set " + std::to_string(ids[i]) + " was never set");
   }
   return 0;
}

}  // namespace paml_amd
