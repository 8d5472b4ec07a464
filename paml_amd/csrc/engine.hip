// engine.hip — C-ABI implementation (include/paml_amd.h) over the HIP kernels in kernels.h.
// Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC engine.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <memory>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is dlopen'ed by paml_amd_comm_* (see Rccl below)

#include "../../include/paml_amd.h"
#include "jit.h"
#include "kernels.h"
#include "program.h"

using namespace paml_amd;
static_assert(JIT_SCRATCH_BASE == MFMA_RS, "the per-tree kernel addresses the interpreter's overflow-stack scratch");

namespace {

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
      const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
      for (const char *nm : names)
         if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
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
Rccl &rccl()
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
   bool no_pipeline = false, force_gather = false, jit_sync = false, jit_strict = false, valu20 = false, no_fused = false, mfma4 = false, tail = false, no_m20 = false;
   int jit_waves = 0;
   std::string jit_dump, prof_ops;
   int prof_tid = 0;
   bool prof_tiles = false;      // the dump is a workgroup timeline (jit.h proft) instead of per-op stamps
   void read()
   {
      no_pipeline = getenv("PAML_AMD_NO_PIPELINE") != nullptr;
      force_gather = getenv("PAML_AMD_FORCE_GATHER") != nullptr;
      force_stream = getenv("PAML_AMD_FORCE_STREAM") != nullptr;      // experiments: the stream interpreter also on small data sets
      jit_sync = getenv("PAML_AMD_JIT_SYNC") != nullptr;
      jit_strict = getenv("PAML_AMD_JIT_STRICT") != nullptr;
      valu20 = getenv("PAML_AMD_VALU20") != nullptr;
      no_fused = getenv("PAML_AMD_NO_FUSED") != nullptr;
      mfma4 = getenv("PAML_AMD_MFMA4") != nullptr;
      no_m20 = getenv("PAML_AMD_NO_M20") != nullptr;
      tail = getenv("PAML_AMD_TAIL") != nullptr;
      if (const char *v = getenv("PAML_AMD_JIT_WAVES")) jit_waves = atoi(v);        // experiment: the last workgroup forms the total instead of a stage-2 launch
      if (const char *v = getenv("PAML_AMD_JIT_DUMP")) jit_dump = v;
      if (const char *v = getenv("PAML_AMD_PROF_OPS")) prof_ops = v;
      if (const char *v = getenv("PAML_AMD_PROF_TID")) prof_tid = atoi(v);
      prof_tiles = getenv("PAML_AMD_PROF_TILES") != nullptr;
   }
};

}  // namespace

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
   DevBuf<unsigned int> d_zpm;        // fused 4 / 5-state kernel: tip codes pattern-major
   int zpm_words = 0;
   DevBuf<int> d_red_counter;         // "last workgroup adds up the partial sums" tickets, one per batch element
   long bpart_rows = 0; int bpart_cols = 0;      // shape of the last eval_branch's partial-sum array
   double *h_out = nullptr;           // pinned, device-visible: the synchronous entry points have lnL written straight to the host
   size_t h_out_cap = 0;
   bool fused = false;                // the selected kernel forms the reduction itself
   bool fused_mfma4 = false;
   bool rate_per_gene = false;      // paml_amd_set_gene_class_rates: class rates [n_genes][K]
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
   } bl;
   DevBuf<double> d_bl_partials, d_bl_scalef, d_bl_frag;
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
   bool jit_enabled = false, use_jit = false;
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
   DevBuf<double> d2_rowmajor, d2_pint, d2_ptip, d2_pcol, d2_branch, d2_gene_rate;
   bool jit_forced = false;  // asked for by flag / environment (as opposed to switched on by the problem's size)
   // a large tree's kernel takes many seconds to compile: that happens on a worker thread while the interpreter kernels
   // serve the evaluations, and the engine changes over when the code object is there
   struct JitJob {
      std::thread th;
      std::atomic<int> state{0};      // 0 idle, 1 compiling, 2 code ready, 3 failed
      std::string key, src, log;
      std::vector<char> code;
   };
   std::unique_ptr<JitJob> jit_job;
   std::string jit_failed_key;

   std::vector<EigenHost> eigen;
   DevBuf<EigenDev> d_eigen;
   bool eigen_dirty = true;

   int mode = PAML_AMD_MODE_LFUN, K = 1, n_labels = 1;

   // per-evaluation buffers
   DevBuf<double> d_b_qfactor, d_b_freqK, d_b_rate;
   DevBuf<double> d_beb_f, d_beb_part, d_beb_g, d_beb_out, d_beb_pcl;      // BEB grid integral
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
      if (comm && rccl().CommDestroy) (void)rccl().CommDestroy(comm);
      if (h_out) (void)hipHostFree(h_out);
      if (d_prof && env.prof_tiles && prof_words) {      // the last launch's workgroup timeline
         std::vector<unsigned long long> hp(prof_words);
         if (hipMemcpy(hp.data(), d_prof, prof_words * 8, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *f = fopen(env.prof_ops.c_str(), "wb")) {
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
      for (auto ev : ev_pool) (void)hipEventDestroy(ev);
      for (auto ev : ev_used) (void)hipEventDestroy(ev);
      DevBuf<unsigned char> *b1[] = {&d_z, &d_chara_map, &d_is_leaf, &d_ztiles};
      for (auto b : b1) b->release();
      DevBuf<int> *b2[] = {&d_n_chara, &d_gene_off, &d_label, &d_eigen_of, &d_b_eigen_of, &d_beb_iw};
      for (auto b : b2) b->release();
      d_tiles.release();
      d_tiles_full.release();
      d_zpm.release();
      d_red_counter.release();
      d_bl_partials.release(); d_bl_scalef.release(); d_bl_frag.release(); d_code_mask.release();
      d_ops.release();
      d_ops_tmp.release();
      d_label_eff.release();
      d_stream.release();
      d_eigen.release();
      d_pi_plain.release();
      DevBuf<double> *b3[] = {&d_weights, &d_pi, &d_freqK, &d_rate, &d_qfactor, &d_branch, &d_gene_rate, &d_rowmajor,
                              &d_pint, &d_ptip, &d_pcol, &d_fhK, &d_fscale, &d_lnf, &d_b_qfactor, &d_b_freqK, &d_b_rate, &d_beb_f, &d_beb_part, &d_beb_g, &d_beb_out, &d_beb_pcl, &d_partial, &d_out, &d_partials, &d_scalef, &d_stack, &d2_rowmajor, &d2_pint, &d2_ptip, &d2_pcol, &d2_branch, &d2_gene_rate, &d_partial_tot, &d_btot,
                              &d_expA, &d_expB, &d_expSA, &d_expSB, &d_deriv, &d_tt, &d_bpartial, &d_bout};
      for (auto b : b3) b->release();
   }
};

namespace {

int fail(paml_amd_engine *e, int code, const std::string &msg)
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
hipError_t upload(DevBuf<T> &b, const T *src, size_t n, hipStream_t s)
{
   hipError_t r = b.ensure(n);
   if (r != hipSuccess) return r;
   if (n == 0) return hipSuccess;
   // pageable source: the runtime stages the copy, so the host buffer may be reused on return
   return hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, s);
}

int tipw(const paml_amd_engine *e) { return e->kk == KK_MFMA64 ? 64 : e->n; }
// doubles per tip table: mfma64 pads tables of <= 64 codes to one 32 KB stream block
size_t tip_words(const paml_amd_engine *e)
{
   if (e->kk != KK_MFMA64) return (size_t)e->n_codes * e->n;
   return e->n_codes <= 64 ? 4096 : (size_t)e->n_codes * 64;
}
int pint_words(const paml_amd_engine *e) { return e->kk == KK_MFMA64 ? 4096 : e->n * e->n; }

hipEvent_t get_event(paml_amd_engine *e)
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

void mark_on(paml_amd_engine *e, hipStream_t s)
{
   if (!e->profiling) return;
   hipEvent_t ev = get_event(e);
   if (ev) (void)hipEventRecord(ev, s);
}
void mark(paml_amd_engine *e) { mark_on(e, e->stream); }

// Pinned host memory the kernels can write: the synchronous entry points get their scalars without a device-to-host copy.
int ensure_hout(paml_amd_engine *e, size_t n)
{
   if (n <= e->h_out_cap) return 0;
   if (e->h_out) (void)hipHostFree(e->h_out);
   e->h_out = nullptr; e->h_out_cap = 0;
   HIPCHK(hipHostMalloc((void **)&e->h_out, std::max<size_t>(n, 64) * sizeof(double), hipHostMallocDefault));
   e->h_out_cap = std::max<size_t>(n, 64);
   return 0;
}

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
   if (e->kk == KK_MFMA64 && e->tile_patt >= 128 && e->d_z.p && e->d_weights.p) {   // code blocks of the specialised kernel
      e->zt_bytes = jit_zpieces(e->n_tips, e->tile_patt) * 2048;
      HIPCHK(e->d_ztiles.ensure((size_t)e->n_tiles * e->zt_bytes));
      hipLaunchKernelGGL(ztile_kernel, dim3(e->n_tiles), dim3(e->tile_patt), 0, e->stream, e->d_tiles.p, e->d_gene_off.p, e->d_z.p, (long)e->n_patt,
                         e->d_weights.p, e->n_tips, e->zt_bytes, e->d_ztiles.p);
   }
   HIPCHK(hipStreamSynchronize(e->stream));
   return 0;
}

// The one-pattern-per-lane interpreter: the register-stack instantiation that fits the program, else the scratch one.
void launch_valu(paml_amd_engine *e, int max_stack, int n_blocks, const PruneArgs &pr)
{
   const dim3 g(n_blocks), b(256);
   switch (e->kk) {
   case KK_VALU4:
      if (max_stack <= 4) hipLaunchKernelGGL((prune_valu<4, 4, true>), g, b, 0, e->stream, pr);
      else hipLaunchKernelGGL((prune_valu<4, VALU_MAXD_SMALL>), g, b, 0, e->stream, pr);
      break;
   case KK_VALU5:
      if (max_stack <= 4) hipLaunchKernelGGL((prune_valu<5, 4, true>), g, b, 0, e->stream, pr);
      else hipLaunchKernelGGL((prune_valu<5, VALU_MAXD_SMALL>), g, b, 0, e->stream, pr);
      break;
   default:      // 20 states: a register stack costs > 256 VGPRs (one wave per SIMD) and measures 2x slower than scratch
      hipLaunchKernelGGL((prune_valu<20, VALU_MAXD_20>), g, b, 0, e->stream, pr);
      break;
   }
}

// The specialised kernel for `key`: reuse the loaded module or generate + compile + load it.  A compile failure is
// not fatal (the interpreter kernels take over) unless PAML_AMD_JIT_STRICT is set.
template <class GEN>
int ensure_jit(paml_amd_engine *e, const std::string &key, GEN gen, bool *ok)
{
   *ok = false;
   if (e->jit.fn && e->jit.key == key) { *ok = true; return 0; }
   if (e->jit.mod) (void)hipModuleUnload(e->jit.mod);
   e->jit = JitKernel();
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

// Batched evaluations: B parameter sets run as K*B classes of one launch (class index = b*K + iclass), so the pruning
// kernels are unchanged; P(t) and the reduction index the per-element inputs.  Null members = shared set_classes values.
struct BatchSpec {
   int B;
   const int *eigen_of;      // [B][n_genes][K][n_labels]
   const double *qfactor;    // [B][K][n_labels]
   const double *freqK;      // [B][K]
   const double *rate;       // [B][K]
};

int launch_eval(paml_amd_engine *e, const double *branch, const double *gene_rate, const unsigned char *clean,
                double *d_lnL_out, bool want_lnf, const BatchSpec *bs = nullptr, bool want_pipe = false, bool want_fhk = true)
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
   // (worth its event traffic only where the pruning kernel is long: the 21..64-state kernels on >= 10^5 pattern-classes)
   want_pipe = want_pipe && e->kk == KK_MFMA64 && (long)e->n_patt * e->K >= 100000;
   const bool pipe = want_pipe && e->pipe_ok && !bs && !clean && !keep && !new_prog && !e->eigen_dirty && !e->env.no_pipeline;
   if (want_pipe && !e->s2) {
      HIPCHK(hipStreamCreateWithFlags(&e->s2, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&e->ev_entry[0], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e->ev_entry[1], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e->ev_pmat, hipEventDisableTiming));
   }
   hipStream_t ps = e->stream;                    // stream of the uploads and of the P(t) kernel
   if (pipe) {
      // the side stream may overwrite the other P set once everything the main stream held in front of the PREVIOUS pruning
      // kernel is done: that set's last reader (the kernel before it), and the previous evaluation's own uploads and P(t)
      ps = e->s2;
      if (e->have_prev_entry) HIPCHK(hipStreamWaitEvent(e->s2, e->ev_entry[e->entry_sel ^ 1], 0));
   }
   std::vector<EigenDev> tab;
   if (e->eigen_dirty) {
      tab.resize(e->eigen.size());
      for (size_t i = 0; i < e->eigen.size(); i++) {
         const EigenHost &h = e->eigen[i];
         if (h.kind < 0) return fail(e, PAML_AMD_EINVAL, "eigen set " + std::to_string(i) + " was never set");
         tab[i] = EigenDev{h.kind, h.nR, h.kappa, h.U.p, h.V.p, h.Root.p, h.Cijk.p};
      }
   }

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
            for (size_t i = 0; i < cnt; i++)
               if (bs->eigen_of[i] < 0 || bs->eigen_of[i] >= (int)e->eigen.size())
                  return fail(e, PAML_AMD_EINVAL, "eval_batch: eigen_of entry out of range");
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
      std::swap(e->d_rowmajor, e->d2_rowmajor); std::swap(e->d_pint, e->d2_pint); std::swap(e->d_ptip, e->d2_ptip); std::swap(e->d_pcol, e->d2_pcol);
   }
   HIPCHK(e->d_rowmajor.ensure((size_t)psets * nn * n * n));
   if (e->kk == KK_MFMA64) HIPCHK(e->d_pint.ensure((size_t)psets * nn * 4096));
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
      if (e->jit_enabled && !e->env.force_gather && jit_supported(e->prog, e->n_tips, e->n_codes, e->n_pi, 6, jw * 16)) {
         const std::string key = "m" + std::to_string(n) + "c" + std::to_string(e->n_codes) + "w" + std::to_string(jw) + ":" + jit_program_key(e->prog, e->n_tips);
         const bool background = e->prog.ops.size() > 120 &&      /* (roughly: more than 60 taxa, more than 3 s of compilation) */ !e->jit_forced && !e->env.jit_sync && !(e->jit.fn && e->jit.key == key);
         if (!background) {
            int r = ensure_jit(e, key, [&]() { return jit_generate(e->prog, e->n_tips, n, e->n_codes, jw); }, &jit_ok);
            if (r) return r;
         }
         else {
            paml_amd_engine::JitJob *job = e->jit_job.get();
            if (job && job->state.load() >= 2 && job->th.joinable()) job->th.join();
            if (job && job->state.load() == 2 && job->key == key) {          // the code object is there: load it and change over
               if (e->jit.mod) (void)hipModuleUnload(e->jit.mod);
               e->jit = JitKernel();
               if (hipModuleLoadData(&e->jit.mod, job->code.data()) == hipSuccess && hipModuleGetFunction(&e->jit.fn, e->jit.mod, "prune_jit") == hipSuccess) {
                  e->jit.key = key;
                  jit_ok = true;
               }
               else e->jit_failed_key = key;
               e->jit_job.reset();
            }
            else if (job && job->state.load() >= 2) {                        // failed, or compiled for another tree
               if (job->state.load() == 3 && job->key == key) { e->jit_failed_key = key; e->err = "jit: " + job->log; }
               e->jit_job.reset();
               job = nullptr;
            }
            if (!jit_ok && !e->jit_job && e->jit_failed_key != key) {
               e->jit_job.reset(new paml_amd_engine::JitJob());
               job = e->jit_job.get();
               job->key = key;
               job->src = jit_generate(e->prog, e->n_tips, n, e->n_codes, jw);
               job->state.store(1);
               job->th = std::thread([job]() { job->state.store(jit_compile_code(job->src, &job->code, &job->log) == 0 ? 2 : 3); });
            }
         }
      }
      e->use_jit = jit_ok;
      const bool big_tiles = jit_ok || lean;
      const int want_waves = jit_ok ? jw : (lean ? DMA_WAVES : GATHER_WAVES);
      if (big_tiles != e->mfma_dma || want_waves != e->mfma_waves) {
         e->mfma_dma = big_tiles;
         e->mfma_waves = want_waves;
         e->tile_patt = e->mfma_waves * 16;
         int r = build_tiles(e);
         if (r) return r;
         e->partials_valid = false;
         if (clean) return fail(e, PAML_AMD_EINVAL, "eval_dirty: kernel layout changed; run a full evaluation first");
      }
   }
   if (e->kk != KK_MFMA64) {      // 4 / 5 / 20 states: the interpreter unrolled for this tree
      bool jit_ok = false, fused = false;
      // (20 states: the unrolled walk needs > 256 VGPRs and runs at one wave per SIMD, slower than the interpreter)
      if (e->jit_enabled && !keep && n <= 5 && jit_valu_supported(e->prog)) {
         // the fused form (classes inside, LDS tip tables, reduction in the epilogue) when the model fits it
         const ValuFusedPlan pl = jit_valu_fused_plan(e->prog, n, e->n_tips, e->n_codes, Km, e->chunk);
         if (pl.ok && G == 1 && e->n_pi == 1 && e->d_zpm.p && !e->env.no_fused) {
            // 4 states: the matrix-core form (v_mfma_f64_4x4x4) is an experiment kept behind PAML_AMD_MFMA4=1 — same issue slots as
            // the FMA form (an FP64 MFMA of 256 MACs takes 16 cycles, sixteen v_fma_f64 of a wave 64) and four times the
            // integer work per pattern (a lane is a (state, pattern) pair): 0.32 of peak against 0.64, profiles/r02_valu_fused_shapes.txt
            const bool m4 = n == 4 && e->env.mfma4;
            int r = ensure_jit(e, std::string(m4 ? "m4" : "vf") + std::to_string(n) + "c" + std::to_string(e->n_codes) + "k" + std::to_string(Km) + "r" + std::to_string(pl.R) + "w" +
                                     std::to_string(pl.CW) + (pl.cherry ? "y:" : "n:") + jit_program_key(e->prog, e->n_tips),
                               [&]() { return m4 ? jit_generate_mfma4(e->prog, e->n_tips, e->n_codes, Km, e->chunk)
                                                 : jit_generate_valu_fused(e->prog, n, e->n_tips, e->n_codes, Km, e->chunk); }, &jit_ok);
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
         int r = ensure_jit(e, "m20c" + std::to_string(e->n_codes) + ":" + jit_program_key(e->prog, e->n_tips), [&]() { return jit_generate_m20(e->prog, e->n_tips, e->n_codes); }, &jit_ok);
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
      size_t words = e->kk == KK_MFMA64 ? (size_t)K * n_int * e->n_tiles * e->mfma_waves * 1024
                                        : (size_t)K * n_int * e->n_patt * n;
      HIPCHK(e->d_partials.ensure(words));
      HIPCHK(e->d_scalef.ensure((size_t)K * std::max(1, e->tree.n_scale) * e->n_patt));
   }
   int overflow = 0;
   if (e->kk == KK_MFMA64 && e->prog.max_stack > MFMA_RS) {
      overflow = e->prog.max_stack - MFMA_RS;
      HIPCHK(e->d_stack.ensure((size_t)n_blocks * overflow * e->mfma_waves * 1024));
   }

   // Kernel A: batched P(t)
   PmatArgs pa{};
   pa.n = n; pa.n_nodes = nn; pa.root = e->tree.root; pa.K = Km; pa.n_genes = G; pa.n_labels = e->n_labels;
   pa.n_codes = e->n_codes; pa.layout = e->kk == KK_MFMA64 ? 1 : ((e->kk == KK_VALU20 && e->use_jit && e->m20) ? 2 : 0);
   pa.label = e->d_label.p; pa.is_leaf = e->d_is_leaf.p; pa.branch = pipe ? e->d2_branch.p : e->d_branch.p; pa.rate = e->d_rate.p;
   pa.gene_rate = pipe ? e->d2_gene_rate.p : e->d_gene_rate.p; pa.eigen_of = e->d_eigen_of.p; pa.qfactor = e->d_qfactor.p;
   pa.eigen = e->d_eigen.p; pa.n_chara = e->d_n_chara.p; pa.chara_map = e->d_chara_map.p;
   pa.rowmajor = e->d_rowmajor.p; pa.pint = e->d_pint.p; pa.ptip = e->d_ptip.p; pa.tip_words = (long)tip_words(e);
   pa.B = B; pa.branch_bs = nn; pa.gene_rate_bs = G; pa.pcol = e->kk == KK_MFMA64 ? e->d_pcol.p : nullptr;
   if (bs && bs->eigen_of) { pa.eigen_of = e->d_b_eigen_of.p; pa.eigen_of_bs = (long)G * Km * e->n_labels; }
   if (bs && bs->qfactor) { pa.qfactor = e->d_b_qfactor.p; pa.qfactor_bs = (long)Km * e->n_labels; }
   pa.rate_gs = e->rate_per_gene ? Km : 0;
   if (bs && bs->rate) { pa.rate = e->d_b_rate.p; pa.rate_bs = e->rate_per_gene ? (long)G * Km : Km; }
   mark_on(e, ps);
   bool small_pmat = e->kk != KK_MFMA64 && n <= 5;
   for (const EigenHost &h : e->eigen) small_pmat = small_pmat && h.kind != PAML_AMD_EIGEN_QMAT;
   if (small_pmat) hipLaunchKernelGGL(pmat_small_kernel, dim3((nn * psets + 7) / 8), dim3(256), 0, ps, pa, iv);
   else hipLaunchKernelGGL(pmat_kernel, dim3(nn, psets), dim3(256), 2 * 4096 * sizeof(double), ps, pa, iv);
   mark_on(e, ps);
   if (pipe) {      // the pruning kernel (main stream) starts when this P(t) is there
      HIPCHK(hipEventRecord(e->ev_pmat, e->s2));
      HIPCHK(hipStreamWaitEvent(e->stream, e->ev_pmat, 0));
   }
   if (want_pipe) {      // "everything on the main stream in front of this pruning kernel": what the next pipelined evaluation waits for
      HIPCHK(hipEventRecord(e->ev_entry[e->entry_sel], e->stream));
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
   pr.fhK = e->d_fhK.p; pr.partials = e->d_partials.p; pr.scalef = e->d_scalef.p; pr.stack_scratch = e->d_stack.p;
   pr.stack_overflow_slots = overflow; pr.first_matmul = e->prog.first_matmul; pr.n_int = n_int;
   pr.first_tip = e->prog.first_tip;
   pr.stream = e->d_stream.p; pr.n_stream = (int)(e->prog.stream.size() / 2); pr.tip_words = (long)tip_words(e);
   // the reduction's geometry (the fused kernels form the partial sums themselves; the others leave them to reduce_stage1)
   const int chunk = e->chunk, nbg = e->nb_global;
   const int nb = (e->n_patt + chunk - 1) / chunk;
   if ((size_t)nbg * B > e->d_partial.cap) {
      HIPCHK(e->d_partial.ensure((size_t)nbg * B));
      HIPCHK(hipMemsetAsync(e->d_partial.p, 0, e->d_partial.cap * sizeof(double), e->stream));
   }
   if ((size_t)B * RED_TICKET_WORDS > e->d_red_counter.cap) {
      HIPCHK(e->d_red_counter.ensure((size_t)std::max(B, 64) * RED_TICKET_WORDS));
      HIPCHK(hipMemsetAsync(e->d_red_counter.p, 0, e->d_red_counter.cap * sizeof(int), e->stream));
   }
   HIPCHK(e->d_out.ensure(B));
   if (want_lnf) HIPCHK(e->d_lnf.ensure((size_t)B * e->n_patt));
   double *const lnl_out = d_lnL_out ? d_lnL_out : e->d_out.p;
   const bool fused = e->kk != KK_MFMA64 && e->use_jit && e->fused;
   pr.zpm = e->d_zpm.p; pr.zpm_words = e->zpm_words;
   if (fused) {
      pr.zpm = e->d_zpm.p; pr.zpm_words = e->zpm_words; pr.Km = Km; pr.chunk = chunk; pr.first_chunk = e->first_chunk; pr.nb_stride = nbg;
      pr.want_fhk = (want_fhk || e->tree.n_scale) ? 1 : 0;
      pr.freqK = (bs && bs->freqK) ? e->d_b_freqK.p : e->d_freqK.p; pr.freqK_bs = (bs && bs->freqK) ? Km : 0;
      pr.lnf = want_lnf ? e->d_lnf.p : nullptr;
      pr.red_partial = e->d_partial.p; pr.red_out = lnl_out;
      // the total: a one-block stage-2 launch (default), or PAML_AMD_TAIL=1: the workgroup that finishes last forms it (tickets)
      pr.red_counter = (e->comm || !e->env.tail) ? nullptr : e->d_red_counter.p;
   }
   const int prof_stride = std::max((int)e->prog.ops.size() + 3, e->env.prof_tiles ? 96 : 0);      // experiments only
   if (!e->env.prof_ops.empty()) {
      // per-op stamps: a fresh buffer and a dump after every launch; the workgroup timeline: one buffer, overwritten by every
      // launch and written out when the engine goes (nothing between the launches, so that the clock is the production clock)
      const size_t words = (size_t)3 * n_blocks * prof_stride;
      if (!e->env.prof_tiles || words != e->prof_words) {
         if (e->d_prof) (void)hipFree(e->d_prof);
         e->d_prof = nullptr;
         HIPCHK(hipMalloc((void **)&e->d_prof, words * 8));
         HIPCHK(hipMemsetAsync(e->d_prof, 0, words * 8, e->stream));
         e->prof_words = words; e->prof_blocks = n_blocks; e->prof_stride = prof_stride;
      }
      pr.prof = e->d_prof;
      pr.prof_stride = prof_stride;
      pr.prof_tid = e->env.prof_tid;
   }
   mark(e);
   switch (e->kk) {
   case KK_MFMA64:
      if (e->use_jit) {
         void *params[] = {&pr};
         const int grid = std::min(n_blocks, e->n_cu);     // persistent: one 130 KB-LDS workgroup per CU walks the tiles
         HIPCHK(hipModuleLaunchKernel(e->jit.fn, grid, 1, 1, e->mfma_waves * 64, 1, 1, 0, e->stream, params, nullptr));
      }
      else if (use_dma) {
         const size_t lds = (size_t)4 * 4096 * sizeof(double) + (size_t)e->n_tips * 128;
         if (!e->stream_attr_set) {
            HIPCHK(hipFuncSetAttribute((const void *)prune_mfma64_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            e->stream_attr_set = true;
         }
         hipLaunchKernelGGL(prune_mfma64_stream, dim3(n_blocks), dim3(512), lds, e->stream, pr);
      }
      else
         hipLaunchKernelGGL(prune_mfma64_gather<GATHER_WAVES>, dim3(n_blocks), dim3(GATHER_WAVES * 64), 0, e->stream, pr);
      break;
   case KK_VALU4:
   case KK_VALU5:
   case KK_VALU20:
      if (fused) {
         void *params[] = {&pr};
         HIPCHK(hipModuleLaunchKernel(e->jit.fn, nb, B, 1, e->fused_threads, 1, 1, 0, e->stream, params, nullptr));
      }
      else if (e->use_jit && e->m20) {      // persistent: a multiple of the class count, every workgroup keeps its class's P(t) in LDS
         void *params[] = {&pr};
         const int grid = std::min(std::max(K, e->n_cu / K * K), e->n_tiles * K);
         HIPCHK(hipModuleLaunchKernel(e->jit.fn, std::max(grid / K, 1) * K, 1, 1, 512, 1, 1, 0, e->stream, params, nullptr));
      }
      else if (e->use_jit) {
         void *params[] = {&pr};
         HIPCHK(hipModuleLaunchKernel(e->jit.fn, n_blocks, 1, 1, 256, 1, 1, 0, e->stream, params, nullptr));
      }
      else
         launch_valu(e, e->prog.max_stack, n_blocks, pr);
      break;
   }
   mark(e);
   if (pr.prof && !e->env.prof_tiles) {
      std::vector<unsigned long long> hp((size_t)3 * n_blocks * prof_stride);
      HIPCHK(hipMemcpyAsync(hp.data(), e->d_prof, hp.size() * 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
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
   ra.fhK = e->d_fhK.p; ra.weights = e->d_weights.p; ra.freqK = e->d_freqK.p; ra.lnf = want_lnf ? e->d_lnf.p : nullptr;
   ra.partial = e->d_partial.p; ra.out = lnl_out;
   ra.raw = ((e->kk == KK_MFMA64 && e->use_jit) || (e->kk == KK_VALU20 && e->use_jit && e->m20)) ? 1 : 0; ra.fscale = e->d_fscale.p;
   ra.n_patt = e->n_patt; ra.K = Km; ra.mode = e->mode; ra.n_scale = e->tree.n_scale; ra.chunk = chunk;
   ra.first_chunk = e->first_chunk; ra.nb_stride = nbg;
   // (measured on MI355X, 32 taxa x 10^5 nucleotide patterns: 28.2 us per evaluation with the separate one-block launch against
   //  30.2 with tickets — the agent-scope store + two atomics + coherent reads cross the XCDs' L2s and cost more than a launch)
   const bool tail = !e->comm && e->env.tail;
   ra.counter = tail ? e->d_red_counter.p : nullptr;
   if (bs && bs->freqK) { ra.freqK = e->d_b_freqK.p; ra.freqK_bs = Km; }
   mark(e);
   if (!fused) hipLaunchKernelGGL(reduce_stage1, dim3(nb, B), dim3(256), 0, e->stream, ra);
   if (e->comm) {
      HIPCHK(e->d_partial_tot.ensure((size_t)nbg * B));
      const ncclResult_t nr = rccl().AllReduce(e->d_partial.p, e->d_partial_tot.p, (size_t)nbg * B, ncclDouble, ncclSum, e->comm, e->stream);
      if (nr != ncclSuccess) return fail(e, PAML_AMD_EHIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(nr));
      hipLaunchKernelGGL(reduce_stage2, dim3(B), dim3(256), 0, e->stream, (const double *)e->d_partial_tot.p, nbg, ra.out);
   }
   else if (!tail && nbg > 1) hipLaunchKernelGGL(reduce_stage2, dim3(B), dim3(256), 0, e->stream, (const double *)e->d_partial.p, nbg, ra.out);      // (one block per element: stage 1 wrote the total)
   mark(e);
   HIPCHK(hipGetLastError());
   if (e->profiling) e->prof_evals++;
   e->n_eval++;
   if (keep && !clean) e->partials_valid = true;
   e->pmat_valid = true;
   e->pipe_ok = want_pipe;      // (every other entry point clears it)
   return 0;
}


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
   switch (e->kk) {
   case KK_MFMA64:
      hipLaunchKernelGGL(prune_mfma64_gather<GATHER_WAVES>, dim3(n_blocks), dim3(GATHER_WAVES * 64), 0, e->stream, pr);
      break;
   default: launch_valu(e, prog.max_stack, n_blocks, pr); break;
   }
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
   pa.eigen = e->d_eigen.p; pa.n_chara = e->d_n_chara.p; pa.chara_map = e->d_chara_map.p;
   pa.rowmajor = e->d_rowmajor.p; pa.pint = e->d_pint.p; pa.ptip = e->d_ptip.p; pa.tip_words = (long)tip_words(e);
   pa.B = 1; pa.rate_gs = e->rate_per_gene ? K : 0;
   {
      InlineVec iv;
      iv.n_branch = iv.n_rate = 0;
      hipLaunchKernelGGL(pmat_kernel, dim3(nn, psets), dim3(256), 2 * 4096 * sizeof(double), e->stream, pa, iv);
   }
   e->n_pmat += (long)psets * (nn - 1);
   e->prog_valid = false;      // d_branch / P buffers now hold the re-rooted edge data: the next eval rebuilds
   e->partials_valid = false;
   e->pmat_valid = false;
   return 0;
}

}  // namespace

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
   }
   // 20 states: the specialised MFMA kernel trimmed to 2 row blocks x 5 k-blocks beats the scalar-operand kernel 2-3x; the
   // MFMA interpreters (64 MFMAs per product whatever n) do not, so small or keep-partials engines stay on valu20
   // 20 states: the per-tree kernel on v_mfma_f64_4x4x4 (no padding: 25 block products per 16 patterns) for one gene and trees whose
   // internal branches' P(t) fit in LDS; else the 16x16x4 kernel trimmed to 2 row blocks x 5 k-blocks (2-3x the scalar-operand
   // kernel); the MFMA interpreters (64 MFMAs per product whatever n) do not pay, so small or keep-partials engines stay on valu20
   e->want_m20 = n_states == 20 && e->jit_enabled && !(flags & PAML_AMD_KEEP_PARTIALS) && n_genes == 1 && n_tips <= 49 && !e->env.no_m20 && !e->env.valu20;
   const bool mfma20 = n_states == 20 && !e->want_m20 && e->jit_enabled && !(flags & PAML_AMD_KEEP_PARTIALS) && n_tips <= 95 && !e->env.valu20;
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
   if (e->s2) {
      (void)hipStreamSynchronize(e->s2);
      (void)hipStreamDestroy(e->s2);
      for (hipEvent_t ev : {e->ev_entry[0], e->ev_entry[1], e->ev_pmat}) if (ev) (void)hipEventDestroy(ev);
   }
   delete e;
}

const char *paml_amd_last_error(const paml_amd_engine *e) { return e ? e->err.c_str() : "null engine"; }

const char *paml_amd_kernel_name(const paml_amd_engine *e)
{
   if (!e) return "";
   switch (e->kk) {
   case KK_VALU4: return e->use_jit ? (e->fused && e->fused_mfma4 ? "mfma4_jit" : "valu4_jit") : "valu4";
   case KK_VALU5: return e->use_jit ? "valu5_jit" : "valu5";
   case KK_VALU20: return e->use_jit ? (e->m20 ? "mfma4x20_jit" : "valu20_jit") : "valu20";
   default: return e->use_jit ? "mfma64_jit" : (e->mfma_dma ? "mfma64_stream" : "mfma64_gather");
   }
}

// ---- pattern shards over several GPUs ---------------------------------------------------------------------------------
int paml_amd_device_count(void)
{
   int n = 0;
   return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int paml_amd_set_device(int device) { return hipSetDevice(device) == hipSuccess ? 0 : PAML_AMD_EHIP; }

int paml_amd_shard_bounds(long n_patt_global, int world, int rank, long *first, long *count)
{
   if (n_patt_global < 1 || world < 1 || rank < 0 || rank >= world || !first || !count) return PAML_AMD_EINVAL;
   const long chunk = red_chunk(n_patt_global), nb = (n_patt_global + chunk - 1) / chunk;
   const long c0 = nb * rank / world, c1 = nb * (rank + 1) / world;      // chunks [c0, c1): as even as whole chunks allow
   *first = std::min(n_patt_global, c0 * chunk);
   *count = std::min(n_patt_global, c1 * chunk) - *first;
   return 0;
}

int paml_amd_comm_unique_id(void *id128)
{
   if (!id128) return PAML_AMD_EINVAL;
   static_assert(sizeof(ncclUniqueId) == PAML_AMD_COMM_ID_BYTES, "paml_amd.h states the size of ncclUniqueId");
   if (!rccl().load()) return PAML_AMD_EUNSUPPORTED;
   ncclUniqueId id;
   if (rccl().GetUniqueId(&id) != ncclSuccess) return PAML_AMD_EHIP;
   memcpy(id128, &id, sizeof(id));
   return 0;
}

int paml_amd_comm_init(paml_amd_engine *e, int rank, int world, const void *id128, long n_patt_global, long first_pattern)
{
   if (e) e->pipe_ok = false;
   if (!e || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return fail(e, PAML_AMD_EINVAL, "comm_init: bad arguments");
   if (e->comm) return fail(e, PAML_AMD_EINVAL, "comm_init: the engine already has a communicator");
   const int chunk = red_chunk(n_patt_global);
   if (first_pattern < 0 || first_pattern + e->n_patt > n_patt_global || first_pattern % chunk != 0 ||
       (first_pattern + e->n_patt != n_patt_global && e->n_patt % chunk != 0))
      return fail(e, PAML_AMD_EINVAL, "comm_init: the shard must start and (unless it is the last) end at multiples of " +
                                         std::to_string(chunk) + " patterns (paml_amd_shard_bounds)");
   HIPCHK(hipStreamSynchronize(e->stream));
   if (id128) {      // world == 1 with an id: a one-rank communicator (exercises the collective path on a single GPU)
      if (!rccl().load()) return fail(e, PAML_AMD_EUNSUPPORTED, "comm_init: " + rccl().err);
      ncclUniqueId id;
      memcpy(&id, id128, sizeof(id));
      const ncclResult_t nr = rccl().CommInitRank(&e->comm, world, id, rank);
      if (nr != ncclSuccess) {
         e->comm = nullptr;
         return fail(e, PAML_AMD_EHIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(nr));
      }
   }
   e->rank = rank; e->world = world;
   e->n_patt_global = n_patt_global; e->first_patt = first_pattern;
   e->chunk = chunk;
   e->nb_global = (int)((n_patt_global + chunk - 1) / chunk);
   e->first_chunk = (int)(first_pattern / chunk);
   e->d_partial.release();      // re-zeroed at its new size by the next evaluation
   return 0;
}

int paml_amd_comm_destroy(paml_amd_engine *e)
{
   if (e) e->pipe_ok = false;
   if (!e) return PAML_AMD_EINVAL;
   HIPCHK(hipStreamSynchronize(e->stream));
   if (e->comm) (void)rccl().CommDestroy(e->comm);
   e->comm = nullptr;
   e->rank = 0; e->world = 1; e->n_patt_global = e->n_patt; e->first_patt = 0;
   e->chunk = red_chunk(e->n_patt); e->nb_global = (e->n_patt + e->chunk - 1) / e->chunk; e->first_chunk = 0;
   e->d_partial.release();
   return 0;
}

int paml_amd_get_partial_sums(paml_amd_engine *e, double *out, int cap)
{
   if (e) e->pipe_ok = false;
   if (!e || !out) return fail(e, PAML_AMD_EINVAL, "get_partial_sums: null argument");
   if (e->n_eval == 0 || !e->d_partial.p || cap < e->nb_global) return fail(e, PAML_AMD_EINVAL, "get_partial_sums: nothing evaluated yet, or cap < number of chunks");
   const double *src = e->comm ? e->d_partial_tot.p : e->d_partial.p;
   HIPCHK(hipMemcpyAsync(out, src, (size_t)e->nb_global * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return e->nb_global;
}

int paml_amd_comm_info(const paml_amd_engine *e, int *rank, int *world, long *n_patt_global, long *first_pattern, int *chunk)
{
   if (!e) return PAML_AMD_EINVAL;
   if (rank) *rank = e->rank;
   if (world) *world = e->world;
   if (n_patt_global) *n_patt_global = e->n_patt_global;
   if (first_pattern) *first_pattern = e->first_patt;
   if (chunk) *chunk = e->chunk;
   return 0;
}

int paml_amd_set_stream(paml_amd_engine *e, void *hip_stream)
{
   if (e) e->pipe_ok = false;
   if (!e) return PAML_AMD_EINVAL;
   (void)hipStreamSynchronize(e->stream);
   e->stream = (hipStream_t)hip_stream;
   return 0;
}

int paml_amd_set_tips(paml_amd_engine *e, const unsigned char *z, int cleandata, int n_codes, const int *n_chara,
                      const unsigned char *chara_map, const double *weights, const int *gene_off)
{
   if (e) e->pipe_ok = false;
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
   e->cleandata = cleandata ? 1 : 0;
   e->n_codes = n_codes;
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
      hipLaunchKernelGGL(zpm_kernel, dim3((e->n_patt + 255) / 256), dim3(256), 0, e->stream, e->d_z.p, (long)e->n_patt, e->n_tips, e->n_patt,
                         e->zpm_words, e->d_zpm.p);
   }
   HIPCHK(hipStreamSynchronize(e->stream));
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
   if (e) e->pipe_ok = false;
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
   e->prog_valid = false;
   e->partials_valid = false;
   e->bl.valid = false;
   return 0;
}

int paml_amd_set_pi(paml_amd_engine *e, int n_pi, const double *pi)
{
   if (e) e->pipe_ok = false;
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
   e->partials_valid = false;
   e->bl.valid = false;
   return &e->eigen[set_id];
}

int paml_amd_set_eigen_uvroot(paml_amd_engine *e, int set_id, const double *U, const double *V, const double *Root)
{
   if (e) e->pipe_ok = false;
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

int paml_amd_set_eigen_cijk(paml_amd_engine *e, int set_id, int nR, const double *Cijk, const double *Root)
{
   if (e) e->pipe_ok = false;
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
   if (e) e->pipe_ok = false;
   if (e && e->n != 4) return fail(e, PAML_AMD_EINVAL, "set_eigen_k80: needs 4 states");
   EigenHost *h = eigen_slot(e, set_id);
   if (!h) return fail(e, PAML_AMD_EINVAL, "set_eigen_k80: bad arguments");
   h->kind = PAML_AMD_EIGEN_K80;
   h->kappa = kappa;
   return 0;
}

int paml_amd_set_eigen_jc69like(paml_amd_engine *e, int set_id)
{
   if (e) e->pipe_ok = false;
   EigenHost *h = eigen_slot(e, set_id);
   if (!h) return fail(e, PAML_AMD_EINVAL, "set_eigen_jc69like: bad arguments");
   h->kind = PAML_AMD_EIGEN_JC69LIKE;
   return 0;
}

int paml_amd_set_eigen_qmat(paml_amd_engine *e, int set_id, const double *Q)
{
   if (e) e->pipe_ok = false;
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
   if (e) e->pipe_ok = false;
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
   e->rate_per_gene = false;
   HIPCHK(upload(e->d_qfactor, q.data(), q.size(), e->stream));
   HIPCHK(upload(e->d_eigen_of, eigen_of, (size_t)e->n_genes * K * n_labels, e->stream));
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
   e->pipe_ok = false;
   if (!rate) { e->rate_per_gene = false; return 0; }      // back to the rates of set_classes needs a new set_classes
   HIPCHK(upload(e->d_rate, rate, (size_t)e->n_genes * e->K, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   e->rate_per_gene = true;
   e->partials_valid = false;
   e->bl.valid = false;
   return 0;
}

int paml_amd_eval(paml_amd_engine *e, const double *branch, const double *gene_rate, double *lnL, double *lnf,
                  double *fhK)
{
   if (e) e->pipe_ok = false;
   if (!e || !branch || !lnL) return fail(e, PAML_AMD_EINVAL, "eval: null argument");
   int r = ensure_hout(e, 1);
   if (r) return r;
   r = launch_eval(e, branch, gene_rate, nullptr, e->h_out, lnf != nullptr, nullptr, false, fhK != nullptr);
   if (r) return r;
   if (lnf) HIPCHK(hipMemcpyAsync(lnf, e->d_lnf.p, (size_t)e->n_patt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (fhK)
      HIPCHK(hipMemcpyAsync(fhK, e->d_fhK.p, (size_t)e->K * e->n_patt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   *lnL = e->h_out[0];
   return 0;
}

int paml_amd_eval_batch(paml_amd_engine *e, int n_batch, const double *branch, const double *gene_rate, const int *eigen_of,
                        const double *qfactor, const double *freqK, const double *rate, double *lnL, double *lnf)
{
   if (e) e->pipe_ok = false;
   if (!e || !branch || !lnL || n_batch < 1) return fail(e, PAML_AMD_EINVAL, "eval_batch: bad arguments");
   if ((long)n_batch * e->K * e->n_genes > 65535) return fail(e, PAML_AMD_EINVAL, "eval_batch: n_batch * K * n_genes > 65535");
   BatchSpec bs{n_batch, eigen_of, qfactor, freqK, rate};
   int r = ensure_hout(e, n_batch);
   if (r) return r;
   r = launch_eval(e, branch, gene_rate, nullptr, e->h_out, lnf != nullptr, &bs, false, false);
   if (r) return r;
   if (lnf)
      HIPCHK(hipMemcpyAsync(lnf, e->d_lnf.p, (size_t)n_batch * e->n_patt * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   memcpy(lnL, e->h_out, (size_t)n_batch * sizeof(double));
   return 0;
}

int paml_amd_eval_adg(paml_amd_engine *e, const double *branch, const double *gene_rate, const double *MK, const int *pose, int ls,
                      double *lnL)
{
   if (e) e->pipe_ok = false;
   if (!e || !branch || !MK || !pose || !lnL || ls < 1) return fail(e, PAML_AMD_EINVAL, "eval_adg: bad arguments");
   if (e->mode != PAML_AMD_MODE_LFUNDG) return fail(e, PAML_AMD_EINVAL, "eval_adg: needs the lfundG class mode");
   if (e->world > 1) return fail(e, PAML_AMD_EUNSUPPORTED, "eval_adg: the rate chain runs over the sites in order and does not shard (SURVEY 8e)");
   const int K = e->K, np = e->n_patt;
   for (int i = 0; i < ls; i++)
      if (pose[i] < 0 || pose[i] >= np) return fail(e, PAML_AMD_EINVAL, "eval_adg: pose entry out of range");
   int r = launch_eval(e, branch, gene_rate, nullptr, nullptr, false);      // fx_r on the device
   if (r) return r;
   std::vector<double> fhK((size_t)K * np), w(np), b1(K), b2(K);
   HIPCHK(hipMemcpyAsync(fhK.data(), e->d_fhK.p, fhK.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipMemcpyAsync(w.data(), e->d_weights.p, w.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   // the chain over sites in their original order is sequential: host (treesub.c:7456-7492)
   double l = 0;
   if (e->tree.n_scale)
      for (int h = 0; h < np; h++) {
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

// shared front half of the two BEB entry points: checks, uploads, and the scale / lnfx / finish kernels
static int beb_front(paml_amd_engine *e, const char *who, int n_grid, int n_cls, const double *pcl, const int *iw, size_t out_per_patt, BebArgs &a)
{
   if (e->mode != PAML_AMD_MODE_LFUNDG || e->n_eval == 0 || !e->d_fhK.p)
      return fail(e, PAML_AMD_EINVAL, std::string(who) + ": needs a previous evaluation in the lfundG class mode");
   const int K = e->K, np = e->n_patt;
   for (long i = 0; i < (long)n_grid * n_cls; i++)
      if (iw[i] < 0 || iw[i] >= K) return fail(e, PAML_AMD_EINVAL, std::string(who) + ": class index out of range");
   a = BebArgs{};
   a.n_patt = np; a.K = K; a.n_grid = n_grid; a.n_cls = n_cls;
   a.log_form = e->tree.n_scale > 0;      // root_value: with scaling nodes fhK = log f + scale factors
   a.patt_per_blk = 4096;
   a.n_pblk = (np + a.patt_per_blk - 1) / a.patt_per_blk;
   HIPCHK(e->d_beb_f.ensure((size_t)K * np));
   HIPCHK(e->d_beb_part.ensure((size_t)n_grid * a.n_pblk));
   HIPCHK(e->d_beb_g.ensure((size_t)2 * n_grid + K + 1));
   HIPCHK(e->d_beb_out.ensure(out_per_patt * np));
   HIPCHK(upload(e->d_beb_pcl, pcl, (size_t)n_grid * n_cls, e->stream));
   HIPCHK(upload(e->d_beb_iw, iw, (size_t)n_grid * n_cls, e->stream));
   a.fhK = e->d_fhK.p; a.weights = e->d_weights.p; a.f = e->d_beb_f.p; a.pcl = e->d_beb_pcl.p; a.iw = e->d_beb_iw.p;
   a.part = e->d_beb_part.p; a.lnfxs = e->d_beb_g.p; a.wg = e->d_beb_g.p + n_grid; a.fx = e->d_beb_g.p + 2 * n_grid;
   a.w_class = e->d_beb_g.p + 2 * n_grid + 1;
   a.pr_last = e->d_beb_out.p; a.mean_w = a.pr_last + np; a.sd_w = a.mean_w + np;
   hipLaunchKernelGGL(beb_scale, dim3((np + 255) / 256), dim3(256), 0, e->stream, a);
   hipLaunchKernelGGL(beb_lnfx, dim3(a.n_pblk, (n_grid + 63) / 64), dim3(256), 0, e->stream, a);
   hipLaunchKernelGGL(beb_finish, dim3(1), dim3(256), 0, e->stream, a);
   return 0;
}

int paml_amd_beb_grid(paml_amd_engine *e, int n_grid, int n_cls, const double *pcl, const int *iw, const double *w_class,
                      double *ln_fx, double *pr_last, double *mean_w, double *sd_w)
{
   if (e) e->pipe_ok = false;
   if (!e || n_grid < 1 || n_cls < 1 || !pcl || !iw || !w_class || !pr_last || !mean_w || !sd_w)
      return fail(e, PAML_AMD_EINVAL, "beb_grid: bad arguments");
   if (e->K > BEB_MAXK) return fail(e, PAML_AMD_EUNSUPPORTED, "beb_grid: more than 32 classes");
   BebArgs a;
   const int np = e->n_patt;
   if (int rc = beb_front(e, "beb_grid", n_grid, n_cls, pcl, iw, 3, a)) return rc;
   HIPCHK(hipMemcpyAsync((double *)a.w_class, w_class, (size_t)e->K * sizeof(double), hipMemcpyHostToDevice, e->stream));
   hipLaunchKernelGGL(beb_post, dim3((np + 255) / 256), dim3(256), 0, e->stream, a);
   HIPCHK(hipGetLastError());
   HIPCHK(hipMemcpyAsync(pr_last, a.pr_last, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipMemcpyAsync(mean_w, a.mean_w, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipMemcpyAsync(sd_w, a.sd_w, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (ln_fx) HIPCHK(hipMemcpyAsync(ln_fx, a.fx, sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return 0;
}

int paml_amd_beb_grid_classes(paml_amd_engine *e, int n_grid, int n_cls, const double *pcl, const int *iw, double *ln_fx, double *post)
{
   if (e) e->pipe_ok = false;
   if (!e || n_grid < 1 || n_cls < 1 || n_cls > BEB_MAXCLS || !pcl || !iw || !post)
      return fail(e, PAML_AMD_EINVAL, "beb_grid_classes: bad arguments (at most 8 mixture classes per grid point)");
   BebArgs a;
   const int np = e->n_patt;
   if (int rc = beb_front(e, "beb_grid_classes", n_grid, n_cls, pcl, iw, (size_t)n_cls, a)) return rc;
   hipLaunchKernelGGL(beb_post_classes, dim3((np + 255) / 256), dim3(256), 0, e->stream, a);
   HIPCHK(hipGetLastError());
   HIPCHK(hipMemcpyAsync(post, a.pr_last, (size_t)n_cls * np * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (ln_fx) HIPCHK(hipMemcpyAsync(ln_fx, a.fx, sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return 0;
}

int paml_amd_eval_device(paml_amd_engine *e, const double *branch, const double *gene_rate, double *d_lnL)
{
   if (!e || !branch || !d_lnL) return fail(e, PAML_AMD_EINVAL, "eval_device: null argument");
   return launch_eval(e, branch, gene_rate, nullptr, d_lnL, false, nullptr, true, false);
}

int paml_amd_eval_dirty(paml_amd_engine *e, const double *branch, const double *gene_rate, const unsigned char *clean,
                        double *lnL)
{
   if (e) e->pipe_ok = false;
   if (!e || !branch || !lnL || !clean) return fail(e, PAML_AMD_EINVAL, "eval_dirty: null argument");
   int r = ensure_hout(e, 1);
   if (r) return r;
   r = launch_eval(e, branch, gene_rate, clean, e->h_out, false);
   if (r) return r;
   HIPCHK(hipStreamSynchronize(e->stream));
   *lnL = e->h_out[0];
   return 0;
}

int paml_amd_eval_branch(paml_amd_engine *e, int node_b, int n_t, const double *t, const double *branch,
                         const double *gene_rate, double *lnL, double *dlnL, double *ddlnL)
{
   if (e) e->pipe_ok = false;
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
   const size_t words = mfma ? (size_t)K * n_int * e->n_tiles_full * GATHER_WAVES * 1024 : (size_t)K * n_int * e->n_patt * n;
   if (words > e->d_bl_partials.cap) { HIPCHK(e->d_bl_partials.ensure(words)); bc.valid = false; }
   const bool scaled = T.n_scale > 0;
   if (scaled && (size_t)K * T.n_scale * e->n_patt > e->d_bl_scalef.cap) { HIPCHK(e->d_bl_scalef.ensure((size_t)K * T.n_scale * e->n_patt)); bc.valid = false; }
   std::vector<double> gr(G, 1.0);
   if (gene_rate) gr.assign(gene_rate, gene_rate + G);
   if (!bc.valid || (int)bc.up.size() != nn || bc.K != K || bc.gr != gr) {
      bc.up.assign(nn, -2); bc.ok.assign(nn, 0); bc.br.assign(nn, -1.0); bc.gr = gr; bc.K = K;
      bc.valid = true;
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
   if (e->eigen_dirty) {
      std::vector<EigenDev> tab(e->eigen.size());
      for (size_t i = 0; i < e->eigen.size(); i++) {
         const EigenHost &h = e->eigen[i];
         if (h.kind < 0) return fail(e, PAML_AMD_EINVAL, "eigen set " + std::to_string(i) + " was never set");
         tab[i] = EigenDev{h.kind, h.nR, h.kappa, h.U.p, h.V.p, h.Root.p, h.Cijk.p};
      }
      HIPCHK(upload(e->d_eigen, tab.data(), tab.size(), st));
      e->eigen_dirty = false;
   }
   HIPCHK(upload(e->d_gene_rate, gr.data(), gr.size(), st));
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
      pa.eigen = e->d_eigen.p; pa.n_chara = e->d_n_chara.p; pa.chara_map = e->d_chara_map.p;
      pa.rowmajor = e->d_rowmajor.p; pa.pint = e->d_pint.p; pa.ptip = e->d_ptip.p; pa.tip_words = (long)tip_words(e);
      pa.B = 1; pa.rate_gs = e->rate_per_gene ? K : 0;
      {
         InlineVec iv;
         iv.n_branch = iv.n_rate = 0;
         hipLaunchKernelGGL(pmat_kernel, dim3(nn, psets), dim3(256), 2 * 4096 * sizeof(double), st, pa, iv);
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
      pr.first_tip = -1; pr.tip_words = (long)tip_words(e);
      if (mfma) hipLaunchKernelGGL(prune_mfma64_gather<GATHER_WAVES>, dim3(n_blocks), dim3(GATHER_WAVES * 64), 0, st, pr);
      else launch_valu(e, prog.max_stack, n_blocks, pr);
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
   e->bpart_rows = nbg; e->bpart_cols = n_out;
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
   if (e->comm) {      // the exchange step of the branch-local evaluation
      const ncclResult_t nr = rccl().AllReduce(e->d_bpartial.p, e->d_bpartial.p, (size_t)nbg * n_out, ncclDouble, ncclSum, e->comm, st);
      if (nr != ncclSuccess) return fail(e, PAML_AMD_EHIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(nr));
   }
   hipLaunchKernelGGL(branch_reduce_kernel, dim3(1), dim3(256), 0, st, (const double *)e->d_bpartial.p, (int)nbg, n_out, e->d_bout.p);
   HIPCHK(hipGetLastError());
   {
      int r = ensure_hout(e, (size_t)n_t * 3);
      if (r) return r;
   }
   HIPCHK(hipMemcpyAsync(e->h_out, e->d_bout.p, (size_t)n_t * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
   HIPCHK(hipStreamSynchronize(st));      // the one host synchronisation of the call
   for (int i = 0; i < n_t; i++) { lnL[i] = e->h_out[3 * i]; dlnL[i] = e->h_out[3 * i + 1]; ddlnL[i] = e->h_out[3 * i + 2]; }
   e->n_branch_eval++;
   return 0;
}

int paml_amd_branch_counters(const paml_amd_engine *e, long *n_calls, long *n_nodes_recomputed)
{
   if (!e) return PAML_AMD_EINVAL;
   if (n_calls) *n_calls = e->n_branch_eval;
   if (n_nodes_recomputed) *n_nodes_recomputed = e->n_branch_nodes;
   return 0;
}

int paml_amd_get_branch_partials(paml_amd_engine *e, double *out, long cap, long *rows, int *cols)
{
   if (!e || !rows || !cols) return PAML_AMD_EINVAL;
   *rows = e->bpart_rows; *cols = e->bpart_cols;
   if (!out) return 0;
   if (cap < e->bpart_rows * e->bpart_cols || !e->d_bpartial.p) return fail(e, PAML_AMD_EINVAL, "get_branch_partials: no branch evaluation yet, or the buffer is too small");
   HIPCHK(hipMemcpy(out, e->d_bpartial.p, (size_t)e->bpart_rows * e->bpart_cols * sizeof(double), hipMemcpyDeviceToHost));
   return 0;
}

int paml_amd_node_posterior(paml_amd_engine *e, int node, const double *branch, const double *gene_rate, double *post)
{
   if (e) e->pipe_ok = false;
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
   return 0;
}

int paml_amd_get_pmat(paml_amd_engine *e, int gene, int iclass, int node, double *P)
{
   if (e) e->pipe_ok = false;
   if (!e || !P || !e->d_rowmajor.p) return fail(e, PAML_AMD_EINVAL, "get_pmat: nothing evaluated yet");
   if (!e->pmat_valid)
      return fail(e, PAML_AMD_EINVAL, "get_pmat: the P(t) buffers hold the re-rooted matrices of eval_branch / node_posterior; run an evaluation first");
   if (gene < 0 || gene >= e->n_genes || iclass < 0 || iclass >= e->K || node < 0 || node >= e->tree.n_nodes ||
       node == e->tree.root)
      return fail(e, PAML_AMD_EINVAL, "get_pmat: index out of range");
   const size_t nn2 = (size_t)e->n * e->n;
   const double *src = e->d_rowmajor.p + ((size_t)(gene * e->K + iclass) * e->tree.n_nodes + node) * nn2;
   HIPCHK(hipMemcpyAsync(P, src, nn2 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return 0;
}

int paml_amd_get_partials(paml_amd_engine *e, int node, int iclass, double *conP)
{
   if (e) e->pipe_ok = false;
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
   const int MW = e->mfma_waves;
   const size_t groups = (size_t)e->n_tiles * MW;
   std::vector<double> raw(groups * 1024);
   const double *src = e->d_partials.p + ((size_t)iclass * n_int + (node - e->n_tips)) * groups * 1024;
   HIPCHK(hipMemcpyAsync(raw.data(), src, raw.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   // native [group][m][lane] -> [h][state]; lane = (state & 3) * 16 + (h & 15), m = state >> 2
   std::vector<int2> tiles;
   for (int g = 0; g < e->n_genes; g++)
      for (int h = e->gene_off[g]; h < e->gene_off[g + 1]; h += e->tile_patt) tiles.push_back(make_int2(g, h));
   for (size_t t = 0; t < tiles.size(); t++) {
      const int hend = e->gene_off[tiles[t].x + 1];
      for (int w = 0; w < MW; w++)
         for (int hl = 0; hl < 16; hl++) {
            const int h = tiles[t].y + w * 16 + hl;
            if (h >= hend) continue;
            const double *grp = raw.data() + (t * MW + w) * 1024;
            for (int j = 0; j < n; j++) conP[(size_t)h * n + j] = grp[(j >> 2) * 64 + (j & 3) * 16 + hl];
         }
   }
   return 0;
}

int paml_amd_get_scale(paml_amd_engine *e, int node, int iclass, double *scale)
{
   if (e) e->pipe_ok = false;
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

int paml_amd_debug_program(int n_tips, int n_nodes, int root, const int *sons_ptr, const int *sons,
                           const unsigned char *scale_node, int keep_partials, const unsigned char *clean,
                           int *ops_out, int cap, int *max_stack)
{
   if (!sons_ptr || !sons || n_nodes <= 0 || root < 0 || root >= n_nodes) return PAML_AMD_EINVAL;
   TreeDesc t;
   t.n_tips = n_tips; t.n_nodes = n_nodes; t.root = root;
   t.sons_ptr.assign(sons_ptr, sons_ptr + n_nodes + 1);
   t.sons.assign(sons, sons + sons_ptr[n_nodes]);
   t.label.assign(n_nodes, 0);
   t.scale_node.assign(n_nodes, 0);
   t.scale_slot.assign(n_nodes, -1);
   if (scale_node)
      for (int i = 0; i < n_nodes; i++)
         if (scale_node[i]) { t.scale_node[i] = 1; t.scale_slot[i] = t.n_scale++; }
   Program p = build_program(t, keep_partials != 0, clean);
   if (max_stack) *max_stack = p.max_stack;
   if (ops_out)
      for (int i = 0; i < (int)p.ops.size() && i < cap; i++) {
         ops_out[4 * i] = p.ops[i].code; ops_out[4 * i + 1] = p.ops[i].a;
         ops_out[4 * i + 2] = p.ops[i].b; ops_out[4 * i + 3] = p.ops[i].c;
      }
   return (int)p.ops.size();
}

int paml_amd_debug_jit(int n_tips, int n_nodes, int root, const int *sons_ptr, const int *sons,
                       const unsigned char *scale_node, char *text_out, int cap, int compile)
{
   if (!sons_ptr || !sons || n_nodes <= 0 || root < 0 || root >= n_nodes) return PAML_AMD_EINVAL;
   TreeDesc t;
   t.n_tips = n_tips; t.n_nodes = n_nodes; t.root = root;
   t.sons_ptr.assign(sons_ptr, sons_ptr + n_nodes + 1);
   t.sons.assign(sons, sons + sons_ptr[n_nodes]);
   t.label.assign(n_nodes, 0);
   t.scale_node.assign(n_nodes, 0);
   t.scale_slot.assign(n_nodes, -1);
   if (scale_node)
      for (int i = 0; i < n_nodes; i++)
         if (scale_node[i]) { t.scale_node[i] = 1; t.scale_slot[i] = t.n_scale++; }
   Program p = build_program(t, false, nullptr);
   const int fusedK = (compile & 2) ? (compile >> 16) & 0xff : 0, fusedNC = (compile >> 24) & 0xff;      // bit 1: the fused 4 / 5-state kernel
   int n_states = (compile >> 8) & 0xff;    // 0: the 61-state kernel; 4 / 5 / 20: the one-pattern-per-lane kernels;
   const int compile_all = compile;         // 64 + n: the MFMA kernel trimmed to n states
   compile &= 1;
   std::string text;
   if (n_states > 64) {
      if (!jit_supported(p, n_tips, 61)) return PAML_AMD_EUNSUPPORTED;
      text = jit_generate(p, n_tips, n_states - 64, n_states - 64);
   }
   else if (fusedK && n_states == 20) {
      if (!jit_m20_supported(p, n_tips, 1)) return PAML_AMD_EUNSUPPORTED;
      text = jit_generate_m20(p, n_tips, fusedNC);
   }
   else if (fusedK) {
      const int chunk = (compile_all >> 2) & 0x3f ? ((compile_all >> 2) & 0x3f) * 256 : 256;      // bits 2..7: reduction chunk / 256
      if (!jit_valu_fused_plan(p, n_states, n_tips, fusedNC, fusedK, chunk).ok) return PAML_AMD_EUNSUPPORTED;
      text = (n_states == 4 && getenv("PAML_AMD_MFMA4")) ? jit_generate_mfma4(p, n_tips, fusedNC, fusedK, chunk)
                                                              : jit_generate_valu_fused(p, n_states, n_tips, fusedNC, fusedK, chunk);
   }
   else if (n_states == 4 || n_states == 5 || n_states == 20) {
      if (!jit_valu_supported(p)) return PAML_AMD_EUNSUPPORTED;
      text = jit_generate_valu(p, n_states);
   }
   else {
      int jw = 8;
      if (const char *v = getenv("PAML_AMD_JIT_WAVES")) if (atoi(v) == 12 && jit_zbuffers(n_tips, 192) == 2) jw = 12;
      if (!jit_supported(p, n_tips, 61, 1, 6, jw * 16)) return PAML_AMD_EUNSUPPORTED;
      text = jit_generate(p, n_tips, 61, 64, jw);
   }
   int rc = (int)text.size();
   if (compile) {
      std::vector<char> code;
      std::string log;
      // PAML_AMD_JIT_SHIP=dir: keep the code object there (the library's read-only lib/jit directory is filled this way at build time)
      if (jit_compile_code(text, &code, &log, getenv("PAML_AMD_JIT_SHIP")) != 0) {
         text = log;
         rc = PAML_AMD_EHIP;
      }
   }
   if (text_out && cap > 0) {
      const size_t ncp = std::min((size_t)cap - 1, text.size());
      memcpy(text_out, text.data(), ncp);
      text_out[ncp] = 0;
   }
   return rc;
}

int paml_amd_jit_prebuild(int n_states, int n_tips, int n_codes, int K, long n_patt_global, int n_nodes, int root, const int *sons_ptr,
                          const int *sons, const unsigned char *scale_node, const char *dir, char *log_out, int log_cap)
{
   if (!sons_ptr || !sons || !dir || n_nodes <= 0 || root < 0 || root >= n_nodes || n_states < 2 || n_states > 64) return PAML_AMD_EINVAL;
   TreeDesc t;
   t.n_tips = n_tips; t.n_nodes = n_nodes; t.root = root;
   t.sons_ptr.assign(sons_ptr, sons_ptr + n_nodes + 1);
   t.sons.assign(sons, sons + sons_ptr[n_nodes]);
   t.label.assign(n_nodes, 0);
   t.scale_node.assign(n_nodes, 0);
   t.scale_slot.assign(n_nodes, -1);
   if (scale_node)
      for (int i = 0; i < n_nodes; i++)
         if (scale_node[i]) { t.scale_node[i] = 1; t.scale_slot[i] = t.n_scale++; }
   const Program p = build_program(t, false, nullptr);
   std::string text;
   // the same choices launch_eval makes for an engine of these sizes
   if (n_states == 20 && jit_m20_supported(p, n_tips, 1)) text = jit_generate_m20(p, n_tips, n_codes);
   else if (n_states <= 5) {
      if (!jit_valu_supported(p)) return PAML_AMD_EUNSUPPORTED;
      const int chunk = red_chunk(n_patt_global);
      text = !jit_valu_fused_plan(p, n_states, n_tips, n_codes, K, chunk).ok ? jit_generate_valu(p, n_states)
             : (n_states == 4 && getenv("PAML_AMD_MFMA4"))                     ? jit_generate_mfma4(p, n_tips, n_codes, K, chunk)
                                                                               : jit_generate_valu_fused(p, n_states, n_tips, n_codes, K, chunk);
   }
   else {
      int jw = 8;
      if (const char *v = getenv("PAML_AMD_JIT_WAVES")) if (atoi(v) == 12 && jit_zbuffers(n_tips, 192) == 2) jw = 12;
      if (!jit_supported(p, n_tips, n_codes, 1, 6, jw * 16)) return PAML_AMD_EUNSUPPORTED;
      text = jit_generate(p, n_tips, n_states, n_codes, jw);
   }
   if (const char *dump = getenv("PAML_AMD_JIT_DUMP")) {
      FILE *f = fopen(dump, "w");
      if (f) { fputs(text.c_str(), f); fclose(f); }
   }
   std::vector<char> code;
   std::string log;
   const int rc = jit_compile_code(text, &code, &log, dir);
   if (log_out && log_cap > 0) { strncpy(log_out, log.c_str(), log_cap - 1); log_out[log_cap - 1] = 0; }
   return rc ? PAML_AMD_EHIP : 0;
}

int paml_amd_counters(const paml_amd_engine *e, long *n_eval, long *n_pmat)
{
   if (!e) return PAML_AMD_EINVAL;
   if (n_eval) *n_eval = e->n_eval;
   if (n_pmat) *n_pmat = e->n_pmat;
   return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Site-pattern compression (PatternWeight treesub.c:1386) on the device; kernels in compress.h.  Stand-alone: no engine.
#include "compress.h"

extern "C" int paml_amd_compress_patterns(int n_seq, int n_sites, int width, const unsigned char *chars, const int *gene, int *n_patt,
                                          int *first_site, double *weights, int *pose)
{
   if (n_seq < 1 || n_sites < 1 || width < 1 || !chars || !n_patt || !first_site || !weights || !pose) return PAML_AMD_EINVAL;
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return PAML_AMD_EHIP;
   const int n = n_sites, n_tiles = (n + CMP_TILE - 1) / CMP_TILE, nb = (n + CMP_THREADS - 1) / CMP_THREADS;
   const size_t row = (size_t)n * width;
   unsigned char *d_chars = nullptr, *d_dig = nullptr;
   int *d_gene = nullptr, *d_idx[2] = {nullptr, nullptr}, *d_hist = nullptr, *d_head = nullptr, *d_sums = nullptr, *d_pose = nullptr, *d_first = nullptr,
       *d_start = nullptr, *d_total = nullptr, *d_rowsum = nullptr;
   double *d_w = nullptr;
   int rc = 0, total = 0, cur = 0;
#define CMPCHK(call) do { if ((call) != hipSuccess) { rc = PAML_AMD_EHIP; goto done; } } while (0)
   CMPCHK(hipMalloc(&d_chars, row * n_seq)); CMPCHK(hipMalloc(&d_dig, (size_t)n));
   CMPCHK(hipMalloc(&d_idx[0], (size_t)n * 4)); CMPCHK(hipMalloc(&d_idx[1], (size_t)n * 4));
   CMPCHK(hipMalloc(&d_hist, (size_t)256 * n_tiles * 4));
   CMPCHK(hipMalloc(&d_head, (size_t)n * 4)); CMPCHK(hipMalloc(&d_sums, (size_t)n_tiles * 4));
   CMPCHK(hipMalloc(&d_pose, (size_t)n * 4)); CMPCHK(hipMalloc(&d_first, (size_t)n * 4)); CMPCHK(hipMalloc(&d_start, (size_t)n * 4));
   CMPCHK(hipMalloc(&d_total, 4)); CMPCHK(hipMalloc(&d_w, (size_t)n * 8)); CMPCHK(hipMalloc(&d_rowsum, 256 * 4));
   CMPCHK(hipMemcpy(d_chars, chars, row * n_seq, hipMemcpyHostToDevice));
   if (gene) { CMPCHK(hipMalloc(&d_gene, (size_t)n * 4)); CMPCHK(hipMemcpy(d_gene, gene, (size_t)n * 4, hipMemcpyHostToDevice)); }
   {
      CompressArgs a{};
      a.n_sites = n; a.n_seq = n_seq; a.width = width; a.n_tiles = n_tiles; a.row_stride = (long)row; a.chars = d_chars; a.gene = d_gene; a.hist = d_hist; a.dig = d_dig;
      hipLaunchKernelGGL(cmp_iota, dim3(nb), dim3(CMP_THREADS), 0, 0, d_idx[0], n);
      // least significant key byte first: the last character of the last sequence ... the first of the first, then the gene
      for (int kb = n_seq * width - 1; kb >= (gene ? -1 : 0); kb--) {
         a.seq = kb < 0 ? -1 : kb / width; a.pos = kb < 0 ? 0 : kb % width;
         a.idx_in = d_idx[cur]; a.idx_out = d_idx[cur ^ 1];
         hipLaunchKernelGGL(cmp_hist, dim3(n_tiles), dim3(CMP_THREADS), 0, 0, a);
         hipLaunchKernelGGL(cmp_row_sums, dim3(256), dim3(CMP_THREADS), 0, 0, d_hist, n_tiles, d_rowsum);
         hipLaunchKernelGGL(cmp_row_scan, dim3(256), dim3(CMP_THREADS), 0, 0, d_hist, n_tiles, d_rowsum);
         hipLaunchKernelGGL(cmp_scatter, dim3(n_tiles), dim3(CMP_THREADS), 0, 0, a);
         cur ^= 1;
      }
      a.idx_in = d_idx[cur];
      hipLaunchKernelGGL(cmp_heads, dim3(nb), dim3(CMP_THREADS), 0, 0, a, d_head);
      hipLaunchKernelGGL(cmp_tile_sums, dim3(n_tiles), dim3(CMP_THREADS), 0, 0, d_head, n, d_sums);
      hipLaunchKernelGGL(cmp_scan1, dim3(1), dim3(1024), 0, 0, d_sums, (long)n_tiles, d_total);
      hipLaunchKernelGGL(cmp_number, dim3(n_tiles), dim3(CMP_THREADS), 0, 0, d_head, d_sums, d_idx[cur], n, d_pose, d_first, d_start);
      CMPCHK(hipGetLastError());
      CMPCHK(hipMemcpy(&total, d_total, 4, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL(cmp_weights, dim3((total + CMP_THREADS - 1) / CMP_THREADS), dim3(CMP_THREADS), 0, 0, d_start, total, n, d_w);
      CMPCHK(hipGetLastError());
   }
   CMPCHK(hipMemcpy(pose, d_pose, (size_t)n * 4, hipMemcpyDeviceToHost));
   CMPCHK(hipMemcpy(first_site, d_first, (size_t)total * 4, hipMemcpyDeviceToHost));
   CMPCHK(hipMemcpy(weights, d_w, (size_t)total * 8, hipMemcpyDeviceToHost));
   *n_patt = total;
done:
#undef CMPCHK
   {
      void *bufs[] = {d_chars, d_gene, d_idx[0], d_idx[1], d_hist, d_head, d_sums, d_pose, d_first, d_start, d_total, d_w, d_rowsum, d_dig};
      for (void *b : bufs) (void)hipFree(b);
   }
   return rc;
}
