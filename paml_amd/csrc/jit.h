// jit.h — per-tree specialised pruning kernel.
//
// The interpreter kernels (kernels.h) pay for their generality on every op: a switch dispatch, scalar
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

#include <algorithm>
#include <cstdio>
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
};

// Which programs the generator covers.
inline bool jit_supported(const Program &p, int n_tips, int n_codes, int max_arrays = 6)
{
   if (n_tips > MFMA_ZT || n_codes > 64 || p.ops.size() > 400) return false;
   for (const Op &o : p.ops)
      if (o.code == OP_STORE || o.code == OP_LOAD) return false;   // keep-partials layouts stay with the interpreter
   return p.max_stack + 2 <= max_arrays;
}

inline std::string jit_program_key(const Program &p, int n_tips)
{
   std::ostringstream k;
   k << n_tips << ":";
   for (const Op &o : p.ops) k << o.code << "," << o.a << "," << o.b << ";";
   return k.str();
}

// Emit the straight-line kernel for one program.
inline std::string jit_generate(const Program &p, int n_tips)
{
   std::ostringstream s;
   const int nblk = (int)p.stream.size() / 2;
   s << "#include \"device_common.h\"\nusing namespace paml_amd;\n";
   s << "extern \"C\" __global__ __launch_bounds__(512, 2) void prune_jit(PruneArgs a)\n{\n";
   s << "   JIT_PROLOGUE(" << n_tips << ")\n";
   int issued = 0;
   auto issue = [&]() {
      const int is_tip = p.stream[2 * issued], node = p.stream[2 * issued + 1];
      s << "   " << (is_tip ? "JIT_ISSUE_T(" : "JIT_ISSUE_P(") << issued << ", " << node << ");\n";
      issued++;
   };
   // what starts a tile: tile variables, first operand blocks, tip codes into registers
   const int n_first = std::min(3, nblk);
   auto emit_first_blocks = [&](std::ostringstream &o) {      // the next tile's first operand blocks
      for (int i = 0; i < n_first; i++) {
         const int is_tip = p.stream[2 * i], node = p.stream[2 * i + 1];
         o << "      " << (is_tip ? "JIT_ISSUE_NT(" : "JIT_ISSUE_NP(") << i << ", " << node << ");\n";
      }
   };
   s << "   JIT_NEXT_SET()\n   {\n";
   emit_first_blocks(s);
   s << "   }\n   JIT_ZLOAD(" << n_tips << ")\n";
   issued = n_first;
   s << "   int ptile = 1;\n   for (;; ptile = 0) {\n";
   s << "   JIT_ADVANCE()\n   JIT_ZSTORE(" << n_tips << ")\n";
   int last_mm = -1;
   for (size_t i = 0; i < p.ops.size(); i++)
      if (p.ops[i].code == OP_MATMUL || p.ops[i].code == OP_MATMUL_POP) last_mm = (int)i;

   // register arrays: a free list; `cur` names the array holding the partial under construction
   const bool fuse_tips = !getenv("PAML_AMD_JIT_NOFUSE");   // cherries gathered under the preceding matmul (one more array)
   const int NA = p.max_stack + 2 + (fuse_tips ? 1 : 0);
   for (int i = 0; i < NA; i++) s << "   v4d A" << i << "[4];\n";
   std::vector<int> freeA;
   for (int i = NA - 1; i >= 0; i--) freeA.push_back(i);
   auto alloc = [&]() { int r = freeA.back(); freeA.pop_back(); return r; };
   auto release = [&](int r) { freeA.push_back(r); };
   std::vector<int> slot(256, -1);   // stack slot -> array
   int cur = -1;
   int consumed = 0;
   const int ZR = (n_tips * 128 + 511) / 512;     // tip-code loads per thread (JIT_ZLOAD)
   int extra_loads = 0;                           // ordinary loads issued after the newest DMA that may stay in flight
   const bool spread = !getenv("PAML_AMD_JIT_NOSPREAD");
   // make the next c blocks visible, then top the ring up — at once, or (defer) as a `side` functor that spreads the
   // refill's pieces over the first `iters` k-block pairs of the matmul that follows
   auto step = [&](int c, bool defer = false, int iters = 8, int now = 1) -> std::string {
      s << "   JIT_WAIT(" << 4 * (issued - (consumed + c)) + extra_loads << "); " << (getenv("PAML_AMD_JIT_NOBAR") ? "" : "__syncthreads();") << "\n";
      if (!defer || !spread) {
         while (issued < consumed + 4 && issued < nblk) issue();
         return "JitNoSide()";
      }
      while (issued < consumed + now && issued < nblk) issue();   // needed within this very step: no delay
      std::vector<std::string> pieces;
      while (issued < consumed + 4 && issued < nblk) {
         const int is_tip = p.stream[2 * issued], node = p.stream[2 * issued + 1];
         for (int c4 = 0; c4 < 4; c4++)
            pieces.push_back(std::string(is_tip ? "JIT_PIECE_T(" : "JIT_PIECE_P(") + std::to_string(issued) + ", " + std::to_string(node) +
                             ", " + std::to_string(c4) + ");");
         issued++;
      }
      if (pieces.empty()) return "JitNoSide()";
      const int per = ((int)pieces.size() + iters - 1) / iters;
      std::string f = "[&](int kb2) {";
      for (size_t i = 0; i < pieces.size(); i += per) {
         f += " if (kb2 == " + std::to_string(i / per) + ") {";
         for (size_t k = i; k < i + per && k < pieces.size(); k++) f += " " + pieces[k];
         f += " }";
      }
      return f + " }";
   };
   auto name = [&](int r) { return "A" + std::to_string(r); };

   const bool prof = getenv("PAML_AMD_PROF_OPS") != nullptr;    // kernel experiments: s_memtime stamp after every op
   if (prof) s << "   if (a.prof && tid == a.prof_tid && ptile) a.prof[(long)blockIdx.x * a.prof_stride] = __builtin_amdgcn_s_memtime();\n";
   for (size_t iop = 0; iop < p.ops.size(); iop++) {
      const Op &o = p.ops[iop];
      if (prof)
         s << "   if (a.prof && tid == a.prof_tid && ptile) a.prof[(long)blockIdx.x * a.prof_stride + 1 + " << iop
           << "] = __builtin_amdgcn_s_memtime();\n";
      if ((int)iop == last_mm)     // the next tile's tip codes travel to registers under this tile's last MFMAs
         s << "   work += gridDim.x;\n   JIT_NEXT_SET()\n   JIT_ZLOAD(" << n_tips << ")\n";   // unconditional: static load counts
      switch (o.code) {
      case OP_INIT_ONES:
         if (cur < 0) cur = alloc();
         s << "   jit_init_ones(" << name(cur) << ", q, n);\n";
         break;
      case OP_INIT_TIP:
         if (cur < 0) cur = alloc();
         s << "   jit_init_tip(" << name(cur) << ", JIT_CODE(" << o.a << "), q, a.cleandata);\n";
         break;
      case OP_SET_TIP:
         if (cur < 0) cur = alloc();
         step(1);
         s << "   jit_tip_set(" << name(cur) << ", JIT_BUF(" << consumed << "), JIT_CODE(" << o.a << "), q, lane);\n";
         consumed += 1;
         break;
      case OP_MUL_TIP:
         step(1);
         s << "   jit_tip_mul(" << name(cur) << ", JIT_BUF(" << consumed << "), JIT_CODE(" << o.a << "), q, lane);\n";
         consumed += 1;
         break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2:
         if (cur < 0) cur = alloc();
         step(2);
         s << "   " << (o.code == OP_SET_TIP2 ? "jit_tip2_set(" : "jit_tip2_mul(") << name(cur) << ", JIT_BUF(" << consumed
           << "), JIT_CODE(" << o.a << "), JIT_BUF(" << consumed + 1 << "), JIT_CODE(" << o.b << "), q, lane);\n";
         consumed += 2;
         break;
      case OP_PUSH:
         slot[o.b] = cur;
         cur = -1;
         break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         const int pop = mm_pop_slot(o), push = mm_push_slot(o);
         const int out = alloc();
         if ((int)iop == last_mm) extra_loads = ZR;    // the next tile's tip-code loads were just issued
         // a cherry right after a pushed matmul: its two tip gathers ride under this matmul's second half
         const bool fuse = fuse_tips && push >= 0 && iop + 1 < p.ops.size() && p.ops[iop + 1].code == OP_SET_TIP2;
         const std::string side = step(1, true, fuse ? 4 : 8, fuse ? 3 : 1);
         const int xl = extra_loads;
         extra_loads = 0;
         int tgt = -1;
         if (fuse) {
            const Op &nx = p.ops[iop + 1];
            tgt = alloc();
            s << "   jit_matvec_tip2<" << 4 * (issued - (consumed + 3)) + xl << ">(JIT_BUF(" << consumed << "), lane, " << name(cur) << ", "
              << name(out) << ", JIT_BUF(" << consumed + 1 << "), JIT_CODE(" << nx.a << "), JIT_BUF(" << consumed + 2 << "), JIT_CODE("
              << nx.b << "), q, " << name(tgt) << ", " << side << ");\n";
            consumed += 3;
         }
         else {
            s << "   jit_matvec(JIT_BUF(" << consumed << "), lane, " << name(cur) << ", " << name(out) << ", " << side << ");\n";
            consumed += 1;
         }
         release(cur);
         if (pop >= 0) {
            s << "   jit_mul(" << name(out) << ", " << name(slot[pop]) << ");\n";
            release(slot[pop]);
            slot[pop] = -1;
         }
         if (push >= 0) {
            slot[push] = out;
            cur = -1;
         }
         else
            cur = out;
         if (fuse) cur = tgt;
         if ((int)iop == last_mm) {   // ring is free once every wave has finished this last block
            s << "   if (has_next) {\n      __syncthreads();\n";
            emit_first_blocks(s);
            s << "   }\n";
         }
         if (fuse) {      // the SET_TIP2 is done
            iop++;
            if (prof)
               s << "   if (a.prof && tid == a.prof_tid && ptile) a.prof[(long)blockIdx.x * a.prof_stride + 1 + " << iop
                 << "] = __builtin_amdgcn_s_memtime();\n";
         }
      } break;
      case OP_SCALE:
         s << "   { const double fac = jit_scale(" << name(cur) << ", q, n); lnscale += fac;\n"
           << "     if (a.keep && q == 0 && valid) a.scalef[((long)iclass * a.n_scale + " << o.b << ") * a.n_patt + h] = fac; }\n";
         break;
      case OP_ROOT:
         s << "   jit_root(a, " << name(cur) << ", lnscale, gene, iclass, q, h, valid);\n";
         release(cur);
         cur = -1;
         break;
      default: break;
      }
   }
   // next tile of this persistent workgroup: make sure every wave is done with the ring and the tip codes, then
   // start its operand stream and tip-code loads before looping
   if (last_mm < 0) s << "   work += gridDim.x;\n   JIT_NEXT_SET()\n   JIT_ZLOAD(" << n_tips << ")\n";
   s << "   if (!has_next) break;\n   }\n}\n";
   return s.str();
}

inline std::string jit_source_dir()
{
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
inline int jit_compile_code(const std::string &src, std::vector<char> *code, std::string *log)
{
   hiprtcProgram prog;
   if (hiprtcCreateProgram(&prog, src.c_str(), "prune_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
      *log = "hiprtcCreateProgram failed";
      return -1;
   }
   const std::string inc = "-I" + jit_source_dir();
   const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", inc.c_str()};
   const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
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
