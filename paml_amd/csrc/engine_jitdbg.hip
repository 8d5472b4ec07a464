// engine_jitdbg.hip — the tree-walk program and the per-tree kernel generators without an engine: what the tests inspect, and the
// build-time prebuild of the benchmark configurations' kernels (hiprtc needs no GPU).
// Built for gfx950 only (one of the translation units of libpaml_amd.so, see engine_state.h).
#include "engine_state.h"

// PAML_AMD_PREBUILD_GENES=G (G > 1): the several-genes form of the 4- / 5-state fused kernel and of the 20-state matrix-core kernel
static int prebuild_genes() { const char *v = getenv("PAML_AMD_PREBUILD_GENES"); return v && atoi(v) > 1 ? atoi(v) : 1; }

extern "C" {

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
      if (!jit_m20_supported(p, n_tips, prebuild_genes())) return PAML_AMD_EUNSUPPORTED;
      text = jit_generate_m20(p, n_tips, fusedNC, prebuild_genes());
   }
   else if (fusedK) {
      const int chunk = (compile_all >> 2) & 0x3f ? ((compile_all >> 2) & 0x3f) * 256 : 256;      // bits 2..7: reduction chunk / 256
      if (!jit_valu_fused_plan(p, n_states, n_tips, fusedNC, fusedK, chunk).ok) return PAML_AMD_EUNSUPPORTED;
      text = (n_states == 4 && getenv("PAML_AMD_MFMA4")) ? jit_generate_mfma4(p, n_tips, fusedNC, fusedK, chunk)
                                                              : jit_generate_valu_fused(p, n_states, n_tips, fusedNC, fusedK, chunk, prebuild_genes());
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
   // (PAML_AMD_PREBUILD_KEEP=1: the kernel of a PAML_AMD_KEEP_PARTIALS engine, every internal node's partial stored)
   const Program p = build_program(t, n_states > 20 && getenv("PAML_AMD_PREBUILD_KEEP") != nullptr, nullptr);
   std::string text;
   // the same choices launch_eval makes for an engine of these sizes
   // (PAML_AMD_PREBUILD_COOP=1: the cooperative per-tree kernel of small data sets, whatever the number of states from 20 to 64)
   if (getenv("PAML_AMD_PREBUILD_COOP")) {
      if (!jit_coop_supported(p, n_tips, n_codes)) return PAML_AMD_EUNSUPPORTED;
      text = jit_generate_coop(p, n_tips, n_states);
   }
   else if (n_states == 20 && jit_m20_supported(p, n_tips, prebuild_genes())) text = jit_generate_m20(p, n_tips, n_codes, prebuild_genes());
   else if (n_states <= 5) {
      if (!jit_valu_supported(p)) return PAML_AMD_EUNSUPPORTED;
      const int chunk = red_chunk(n_patt_global);
      text = !jit_valu_fused_plan(p, n_states, n_tips, n_codes, K, chunk).ok ? jit_generate_valu(p, n_states)
             : (n_states == 4 && getenv("PAML_AMD_MFMA4"))                     ? jit_generate_mfma4(p, n_tips, n_codes, K, chunk)
                                                                               : jit_generate_valu_fused(p, n_states, n_tips, n_codes, K, chunk, prebuild_genes());
   }
   else {
      int jw = 8;
      if (const char *v = getenv("PAML_AMD_JIT_WAVES")) if (atoi(v) == 12 && jit_zbuffers(n_tips, 192) == 2) jw = 12;
      if (!jit_supported(p, n_tips, n_codes, 1, 6, jw * 16)) return PAML_AMD_EUNSUPPORTED;
      text = jit_generate(p, n_tips, n_states, n_codes, jw);
      if (!getenv("PAML_AMD_PREBUILD_QUICK")) text = jit_strip_big(text);      // (large trees: the full build, what an engine looks for first)
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

}  // extern "C"
