// program.h — host side: flatten the post-order recursion of ConditionalPNode (codeml.c:3526-3582)
// into a linear program for a small stack machine that every pattern executes in lock-step.
//
// Machine state per pattern: `cur` (the partial being accumulated) and a stack of saved partials.
//   INIT_ONES            cur = 1                       (codeml.c:3539-3540)
//   INIT_TIP  tip        cur = indicator(z[tip][h])    ("young ancestor", codeml.c:3535-3543)
//   MUL_TIP   tip        cur[j] *= sum_{k in code(z[tip][h])} P_tip[j][k]     (codeml.c:3555-3567)
//   PUSH      slot       stack[slot] = cur
//   MATMUL    son        cur = P_son . cur             (cur holds the finished partial of `son`; 3568-3575)
//   MATMUL_POP son slot  cur = stack[slot] * (P_son . cur)
//   SCALE     slot node  NodeScale(node)               (treesub.c:7200-7230)
//   STORE     node       partials[node] = cur          (keep-partials mode)
//   LOAD      node       cur = partials[node]          (clean subtree, com.oldconP)
//   ROOT                 f_h = sum_i pi_i cur[i]       (treesub.c:7728-7729)
//
// Internal sons are evaluated before tip sons and deepest-first (Sethi–Ullman order), so the first
// internal son needs no PUSH and the stack depth is <= log2(n_tips); products commute, so only the
// rounding order differs from the reference's sons[] order.
#pragma once
#include <algorithm>
#include <vector>

#include "device_common.h"

namespace paml_amd {

struct TreeDesc {
   int n_tips = 0, n_nodes = 0, root = -1;
   std::vector<int> sons_ptr, sons, label;
   std::vector<unsigned char> scale_node;
   std::vector<int> scale_slot;    // rank among scaling nodes (treesub.c:7207-7211), -1 if none
   int n_scale = 0;
   bool is_leaf(int i) const { return sons_ptr[i + 1] == sons_ptr[i]; }
};

struct Program {
   std::vector<Op> ops;
   int max_stack = 0;      // stack slots needed
   int n_matmul = 0;
   int first_matmul = -1;  // son of the first MATMUL (-1: none)
   int first_tip = -1;     // first tip whose column table is consumed (-1: none)
   std::vector<int> stream;   // operand blocks in order of use: (is_tip, node) pairs
};

namespace detail {
inline int stack_need(const TreeDesc &t, int node, const unsigned char *clean, std::vector<int> &need)
{
   if (need[node] >= 0) return need[node];
   std::vector<int> kids;
   for (int i = t.sons_ptr[node]; i < t.sons_ptr[node + 1]; i++) {
      int s = t.sons[i];
      if (!t.is_leaf(s)) kids.push_back((clean && clean[s]) ? 0 : stack_need(t, s, clean, need));
   }
   std::sort(kids.begin(), kids.end(), [](int a, int b) { return a > b; });
   int d = 0;
   for (size_t i = 0; i < kids.size(); i++) d = std::max(d, kids[i] + (i > 0 ? 1 : 0));
   return need[node] = d;
}

inline void emit(const TreeDesc &t, int node, const unsigned char *clean, bool keep, std::vector<int> &need,
                 int depth, Program &p)
{
   // internal sons, deepest first
   std::vector<int> kids, tips;
   for (int i = t.sons_ptr[node]; i < t.sons_ptr[node + 1]; i++) {
      int s = t.sons[i];
      (t.is_leaf(s) ? tips : kids).push_back(s);
   }
   std::stable_sort(kids.begin(), kids.end(), [&](int a, int b) {
      int na = (clean && clean[a]) ? 0 : need[a], nb = (clean && clean[b]) ? 0 : need[b];
      return na > nb;
   });
   bool have_cur = false;     // does `cur` already hold factors of this node?
   if (node < t.n_tips) {     // the root is an observed sequence
      p.ops.push_back({OP_INIT_TIP, node, 0, -1});
      have_cur = true;
   }
   for (size_t i = 0; i < kids.size(); i++) {
      int s = kids[i];
      int slot = -1;
      if (have_cur) {
         slot = depth;
         p.ops.push_back({OP_PUSH, node, slot, -1});
         p.max_stack = std::max(p.max_stack, slot + 1);
      }
      if (clean && clean[s])
         p.ops.push_back({OP_LOAD, s, 0, -1});
      else
         emit(t, s, clean, keep, need, have_cur ? depth + 1 : depth, p);
      if (have_cur)
         p.ops.push_back({OP_MATMUL_POP, s, slot + 1, -1});
      else
         p.ops.push_back({OP_MATMUL, s, 0, -1});
      p.n_matmul++;
      have_cur = true;
   }
   if (!have_cur) p.ops.push_back({OP_INIT_ONES, node, 0, -1});
   for (int s : tips) p.ops.push_back({OP_MUL_TIP, s, 0, -1});
   if (!t.scale_node.empty() && t.scale_node[node]) p.ops.push_back({OP_SCALE, node, t.scale_slot[node], -1});
   if (keep && node >= t.n_tips) p.ops.push_back({OP_STORE, node, 0, -1});
}
}  // namespace detail

// What the kernels read beside the ops: prefetch links (see Op), the operand stream, the first MATMUL / tip.  (Also for programs put
// together from several build_program results: the branch-local evaluation's forest of dirty subtrees, engine_branch.hip.)
inline void finish_program(Program &p)
{
   // prefetch links (see Op): walk backwards remembering the next MATMUL's son and the next tip
   {
      int next_mm = -1, next_tip = -1;
      for (int i = (int)p.ops.size() - 1; i >= 0; i--) {
         Op &o = p.ops[i];
         switch (o.code) {
         case OP_MATMUL: case OP_MATMUL_POP: o.c = next_mm; next_mm = o.a; break;
         case OP_MUL_TIP: case OP_SET_TIP: o.c = next_tip; next_tip = o.a; break;
         case OP_SET_TIP2: case OP_MUL_TIP2: o.c = next_tip; next_tip = o.a; break;   // a is consumed first, then b
         default: break;
         }
      }
      p.first_matmul = next_mm;
      p.first_tip = next_tip;
   }
   p.stream.clear();
   for (const Op &o : p.ops) {
      switch (o.code) {
      case OP_MATMUL: case OP_MATMUL_POP: p.stream.push_back(0); p.stream.push_back(o.a); break;
      case OP_MUL_TIP: case OP_SET_TIP: p.stream.push_back(1); p.stream.push_back(o.a); break;
      case OP_SET_TIP2: case OP_MUL_TIP2:
         p.stream.push_back(1); p.stream.push_back(o.a); p.stream.push_back(1); p.stream.push_back(o.b); break;
      default: break;
      }
   }
   // link every MATMUL to the next one so the kernel can prefetch its P while computing
   int next = -1;
   for (int i = (int)p.ops.size() - 1; i >= 0; i--) {
      if (p.ops[i].code == OP_MATMUL || p.ops[i].code == OP_MATMUL_POP) {
         p.ops[i].c = next;
         next = p.ops[i].a;
      }
   }
}

inline Program build_program(const TreeDesc &t, bool keep_partials, const unsigned char *clean)
{
   Program p;
   std::vector<int> need(t.n_nodes, -1);
   detail::stack_need(t, t.root, clean, need);
   detail::emit(t, t.root, clean, keep_partials, need, 0, p);
   p.ops.push_back({OP_ROOT, t.root, 0, -1});
   p.ops.push_back({OP_END, 0, 0, -1});
   // peephole: fuse tip factors so their table gathers are issued together
   {
      std::vector<Op> f;
      const std::vector<Op> &o = p.ops;
      for (size_t i = 0; i < o.size();) {
         const bool mm = o[i].code == OP_MATMUL || o[i].code == OP_MATMUL_POP;
         if (mm && i + 1 < o.size() && o[i + 1].code == OP_PUSH && o[i + 1].b < 255) {
            Op m = o[i];
            m.b |= (o[i + 1].b + 1) << 8;
            f.push_back(m);
            i += 2;
            continue;
         }
         const bool t1 = i + 1 < o.size() && o[i + 1].code == OP_MUL_TIP;
         const bool t2 = i + 2 < o.size() && o[i + 2].code == OP_MUL_TIP;
         if (o[i].code == OP_INIT_ONES && t1 && t2) { f.push_back({OP_SET_TIP2, o[i + 1].a, o[i + 2].a, -1}); i += 3; }
         else if (o[i].code == OP_INIT_ONES && t1) { f.push_back({OP_SET_TIP, o[i + 1].a, 0, -1}); i += 2; }
         else if (o[i].code == OP_MUL_TIP && t1) { f.push_back({OP_MUL_TIP2, o[i].a, o[i + 1].a, -1}); i += 2; }
         else { f.push_back(o[i]); i += 1; }
      }
      p.ops.swap(f);
   }
   finish_program(p);
   return p;
}


}  // namespace paml_amd
