// kernels_prune.h — Kernel B: fused Felsenstein pruning (FP64 MFMA path for 21..64 states, one-pattern-per-lane path for
// 4 / 5 / 20 states) and the two layout kernels that prepare the tip codes for the per-tree kernels.  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

#include "kernel_args.h"

namespace paml_amd {

// ------------------------------------------------------------------------------------------------
// Pruning kernels: arguments shared by the MFMA and VALU variants.
// ------------------------------------------------------------------------------------------------
// ---- mfma64: 21..64 states, FP64 MFMA -----------------------------------------------------------
// One wave owns 16 patterns for the whole tree.  A partial is 16 doubles per lane: lane l holds, for
// pattern (l & 15), the states 4m + (l >> 4), m = 0..15.  That is simultaneously
//   * the B operand of v_mfma_f64_16x16x4_f64 for k-block kb = m  (B[k = l>>4][n = l&15]), and
//   * the D layout of the instruction for row block jb = m>>2, register m&3  (row = (l>>4) + 4 reg),
// so cur' = P . cur chains from node to node entirely in registers: no transposes, no LDS traffic for
// partials.  The A operand (P) is staged once per workgroup per branch into LDS in exactly the order
// lanes consume it (pmat_kernel's `frag` layout), double-buffered so the next branch's P streams in
// under the current MFMAs.  Tip branches are gathers from L2-resident column tables.
// Shared op bodies of the two mfma64 kernels (textual, so every register array keeps static indices).
#define MFMA_EPI_INTO(DST)                                                                                       \
   do {                                                                                                         \
      if (pop < 0) {                                                                                            \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = acc[m >> 2][m & 3];                            \
      }                                                                                                         \
      else if (pop == 0) {                                                                                      \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = s0[m] * acc[m >> 2][m & 3];                    \
      }                                                                                                         \
      else if (pop == 1) {                                                                                      \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = s1[m] * acc[m >> 2][m & 3];                    \
      }                                                                                                         \
      else {                                                                                                    \
         const double *sp2 = a.stack_scratch +                                                                  \
                             (((long)blockIdx.x * a.stack_overflow_slots + (pop - MFMA_RS)) * WAVES + wave) * 1024; \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = sp2[m * 64 + lane] * acc[m >> 2][m & 3];       \
      }                                                                                                         \
   } while (0)

#define MFMA_EPILOGUE()                                                                                          \
   do {                                                                                                         \
      const int pop = mm_pop_slot(op), push = mm_push_slot(op);                                                 \
      if (push < 0) MFMA_EPI_INTO(cur);                                                                         \
      else if (push == 0) MFMA_EPI_INTO(s0);                                                                    \
      else if (push == 1) MFMA_EPI_INTO(s1);                                                                    \
      else {                                                                                                    \
         double tmpv[16];                                                                                       \
         MFMA_EPI_INTO(tmpv);                                                                                   \
         double *sp3 = a.stack_scratch +                                                                        \
                       (((long)blockIdx.x * a.stack_overflow_slots + (push - MFMA_RS)) * WAVES + wave) * 1024;  \
         _Pragma("unroll") for (int m = 0; m < 16; m++) sp3[m * 64 + lane] = tmpv[m];                           \
      }                                                                                                         \
   } while (0)

#define MFMA_CORE_CASES()                                                                                        \
   case OP_INIT_ONES: {                                                                                         \
      _Pragma("unroll") for (int m = 0; m < 16; m++) cur[m] = (4 * m + q < n) ? 1.0 : 0.0;                      \
   } break;                                                                                                     \
   case OP_INIT_TIP: {                                                                                          \
      const int code = TIP_CODE(op.a);                                                                          \
      _Pragma("unroll") for (int m = 0; m < 16; m++) cur[m] = (a.cleandata && 4 * m + q == code) ? 1.0 : 0.0;   \
   } break;

#define MFMA_EXT_CASES()                                                                                         \
   case OP_PUSH: {                                                                                              \
      if (op.b == 0) { _Pragma("unroll") for (int m = 0; m < 16; m++) s0[m] = cur[m]; }                         \
      else if (op.b == 1) { _Pragma("unroll") for (int m = 0; m < 16; m++) s1[m] = cur[m]; }                    \
      else {                                                                                                    \
         double *sp = a.stack_scratch +                                                                         \
                      (((long)blockIdx.x * a.stack_overflow_slots + (op.b - MFMA_RS)) * WAVES + wave) * 1024;   \
         _Pragma("unroll") for (int m = 0; m < 16; m++) sp[m * 64 + lane] = cur[m];                             \
      }                                                                                                         \
   } break;                                                                                                     \
   case OP_SCALE: {                                                                                             \
      double mx = 0;                                                                                            \
      _Pragma("unroll") for (int m = 0; m < 16; m++) mx = cur[m] > mx ? cur[m] : mx;                            \
      double o = __shfl_xor(mx, 16);                                                                            \
      mx = o > mx ? o : mx;                                                                                     \
      o = __shfl_xor(mx, 32);                                                                                   \
      mx = o > mx ? o : mx;                                                                                     \
      double fac;                                                                                               \
      if (mx < 1e-300) {                                                                                        \
         _Pragma("unroll") for (int m = 0; m < 16; m++) cur[m] = (4 * m + q < n) ? 1.0 : 0.0;                   \
         fac = -800;                                                                                            \
      }                                                                                                         \
      else {                                                                                                    \
         _Pragma("unroll") for (int m = 0; m < 16; m++) cur[m] /= mx;                                           \
         fac = log(mx);                                                                                         \
      }                                                                                                         \
      lnscale += fac;                                                                                           \
      if (a.keep && q == 0 && valid) a.scalef[((long)iclass * a.n_scale + op.b) * a.n_patt + h] = fac;          \
   } break;                                                                                                     \
   case OP_STORE: {  /* native layout [class][node][16-pattern group] x part_index(m, lane) (PruneArgs::part_groups) */ \
      double *dst = a.partials + (((long)iclass * a.n_int + (op.a - a.n_tips)) * (long)a.part_groups +          \
                                  ((long)tile * WAVES + wave)) * 1024;                                          \
      part_store(dst, lane, cur);                                                                               \
   } break;                                                                                                     \
   case OP_LOAD: {                                                                                              \
      const double *src = a.partials + (((long)iclass * a.n_int + (op.a - a.n_tips)) * (long)a.part_groups +    \
                                        ((long)tile * WAVES + wave)) * 1024;                                    \
      part_load(src, lane, cur);                                                                                \
   } break;

#define MFMA_ROOT_CASE()                                                                                         \
   case OP_ROOT: {                                                                                              \
      const double *pq = a.pi + (long)(a.n_pi > 1 ? gene : 0) * 64 + q * 16;                                    \
      double f = 0;                                                                                             \
      _Pragma("unroll") for (int m = 0; m < 16; m++) f = fma(pq[m], cur[m], f);                                 \
      f += __shfl_xor(f, 16);                                                                                   \
      f += __shfl_xor(f, 32);                                                                                   \
      if (a.keep && a.n_scale) { /* stored factors summed in slot order (treesub.c:7746-7747) */                \
         lnscale = 0;                                                                                           \
         if (valid)                                                                                             \
            for (int k = 0; k < a.n_scale; k++) lnscale += a.scalef[((long)iclass * a.n_scale + k) * a.n_patt + h]; \
      }                                                                                                         \
      if (q == 0 && valid) {                                                                                    \
         double out = 0;                                                                                        \
         if (a.weights[h] > 0) out = root_value(a, f, lnscale);                                                 \
         a.fhK[(long)iclass * a.n_patt + h] = out;                                                              \
      }                                                                                                         \
   } break;

// register-stack-only epilogue (programs with max_stack <= MFMA_RS)
#define MFMA_EPI_REG(DST)                                                                                        \
   do {                                                                                                         \
      if (pop < 0) {                                                                                            \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = acc[m >> 2][m & 3];                            \
      }                                                                                                         \
      else if (pop == 0) {                                                                                      \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = s0[m] * acc[m >> 2][m & 3];                    \
      }                                                                                                         \
      else {                                                                                                    \
         _Pragma("unroll") for (int m = 0; m < 16; m++) DST[m] = s1[m] * acc[m >> 2][m & 3];                    \
      }                                                                                                         \
   } while (0)
#define MFMA_EPILOGUE_REG()                                                                                      \
   do {                                                                                                         \
      const int pop = mm_pop_slot(op), push = mm_push_slot(op);                                                 \
      if (push < 0) MFMA_EPI_REG(cur);                                                                          \
      else if (push == 0) MFMA_EPI_REG(s0);                                                                     \
      else MFMA_EPI_REG(s1);                                                                                    \
   } while (0)

#ifdef PROF_OPS
#define PROF_STAMP(slot) \
   if (a.prof && tid == a.prof_tid) a.prof[(long)blockIdx.x * a.prof_stride + (slot)] = __builtin_amdgcn_s_memtime()
// sub-stamps inside an op: plane 1 / 2 of the dump (same [block][op] indexing)
#define PROF_SUB(plane, ip) \
   if (a.prof && tid == a.prof_tid) a.prof[((long)(plane)*gridDim.x + blockIdx.x) * a.prof_stride + 1 + (ip)] = __builtin_amdgcn_s_memtime()
#else
#define PROF_STAMP(slot)
#define PROF_SUB(plane, ip)
#endif

// ---- mfma64 "gather": tip columns gathered straight from the L2-resident tables into registers.
// Used for trees with more than MFMA_ZT tips; 4 waves (64 patterns) per workgroup, 2 workgroups per CU.

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void prune_mfma64_gather(PruneArgs a)
{
   __shared__ __attribute__((aligned(16))) double sP[2][4096];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, hl = lane & 15;
   const int tile = blockIdx.x % a.n_tiles, iclass = blockIdx.x / a.n_tiles;
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y;
   const int hend = as_const(a.gene_off)[gene + 1];
   const int h = h0 + wave * 16 + hl;
   const bool valid = h < hend;
   const int hc = valid ? h : hend - 1;
   const long pset = (long)gene * a.K + iclass;
   const double *Pint = a.pint + pset * a.n_nodes * 4096;
   const long tipstride = a.tip_words;
   const double *Ptip = a.ptip + pset * a.n_nodes * tipstride;
   const int n = a.n;

   PROF_STAMP(a.prof_stride - 1);
   if (a.first_matmul >= 0) stage_p<WAVES>(Pint + (long)a.first_matmul * 4096, sP[0], wave, lane);

   double cur[16], s0[16], s1[16];   // every program writes cur/s0/s1 (INIT/SET/PUSH) before reading them
   double lnscale = 0;
   int buf = 0;
#define TIP_CODE(tip) ((int)a.z[(long)(tip)*a.z_stride + hc])
   PROF_STAMP(0);
   const int lane0 = lane;
   for (int ip = 0;; ip++) {
      const Op op = fetch_op(a.ops, ip);
      PROF_STAMP(1 + ip);
      if (op.code == OP_END) break;
      int lane = lane0;             // opaque per-iteration copy: keeps LICM from hoisting (and spilling) lane math
      asm volatile("" : "+v"(lane));
      const int q = lane >> 4;
      switch (op.code) {
         MFMA_CORE_CASES()
         MFMA_EXT_CASES()
         MFMA_ROOT_CASE()
      case OP_EXPORT: {
         if (valid) {
            double *dst = a.export_buf + ((long)iclass * a.n_patt + h) * n;
#pragma unroll
            for (int m = 0; m < 16; m++)
               if (4 * m + q < n) dst[4 * m + q] = cur[m];
            if (a.export_scale && q == 0) a.export_scale[(long)iclass * a.n_patt + h] = lnscale;
         }
      } break;
      case OP_MUL_TIP:
      case OP_SET_TIP: {
         double2 v[8];
         tip_gather(Ptip, tipstride, op.a, TIP_CODE(op.a), q, v);
         if (op.code == OP_SET_TIP) {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] = v[i].x; cur[2 * i + 1] = v[i].y; }
         }
         else {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] *= v[i].x; cur[2 * i + 1] *= v[i].y; }
         }
      } break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2: {
         const int c1 = TIP_CODE(op.a), c2 = TIP_CODE(op.b);
         double2 v[8], w[8];
         tip_gather(Ptip, tipstride, op.a, c1, q, v);
         tip_gather(Ptip, tipstride, op.b, c2, q, w);
         if (op.code == OP_SET_TIP2) {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] = v[i].x * w[i].x; cur[2 * i + 1] = v[i].y * w[i].y; }
         }
         else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
               cur[2 * i] = (cur[2 * i] * v[i].x) * w[i].x;
               cur[2 * i + 1] = (cur[2 * i + 1] * v[i].y) * w[i].y;
            }
         }
      } break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         __syncthreads();   // this branch's P has landed in sP[buf]; every wave is done reading sP[buf^1]
         if (op.c >= 0) stage_p<WAVES>(Pint + (long)op.c * 4096, sP[buf ^ 1], wave, lane);
         v4d acc[4];
         mfma_matvec(sP[buf], lane, cur, acc);
         MFMA_EPILOGUE();
         buf ^= 1;
      } break;
      default: break;
      }
   }
#undef TIP_CODE
}

// ---- mfma64 "coop": small data sets (round 4).  When every 16-pattern group of the data can have a CU to itself, what an evaluation
// costs is the length of ONE wave's walk through the tree — in the gather kernel 64 dependent MFMAs per branch (5 800 cycles with the
// barrier and the operand hand-over; 13 taxa x 79 codon patterns: 39 us per launch, profiles/r04_small_timeline.txt).  Here the four
// waves of a workgroup share ONE group: wave w owns row block w of every product — states 16 w .. 16 w + 15 of every partial, four
// doubles per lane instead of sixteen — so a product is 16 MFMAs per wave; the operand partial is put together in LDS (each wave
// publishes its quarter, one barrier — the one the staged P(t) block needs anyway), tip rows are gathered a quarter per wave, the
// group's character codes come to LDS once: 25 us for the same launch.  Accumulation order per row block (k-blocks ascending into
// one accumulator) and the root sum (wave 0, the gather kernel's own code) are the other kernels': same bits.  Workgroup b: 16-pattern
// group (b & 3) of 64-pattern tile (b >> 2) % n_tiles — the gather kernel's tile table, so the engine changes between the two kernels
// (a batched gradient has many more groups) without rebuilding anything.  No keep-partials ops, at most COOP_SLOTS stack slots.
// (Also built and measured, profiles/r04_small_timeline.txt: the A operands and the next tip step's rows straight from global memory into
// registers two products ahead, no LDS staging of P — slower, 37.7 us: the compiler waits for every outstanding load at each use inside
// the op loop; P(t) in a wave-private ring of three with counted waits and the tip rows as LDS-DMA gathers a product ahead, the
// look-aheads resolved on the host — 3 % faster, same bits, not worth its dozen counted waits.  A dependent MFMA chain costs nothing
// extra (tools/mfma_f64_chain.hip): of a product step's 4 400 cycles the 16 MFMAs are 1 024, the rest is the op loop's serial chain
// of 100 - 300-cycle latencies that one wave per SIMD cannot hide.)
constexpr int COOP_SLOTS = 8;
constexpr int COOP_ZT = 512;      // tips whose codes are staged in LDS (more: read from global memory)

__global__ __launch_bounds__(256) void prune_mfma64_coop(PruneArgs a)
{
   __shared__ __attribute__((aligned(16))) double sP[2][4096];
   __shared__ __attribute__((aligned(16))) double sX[2][1024];      // the operand partial, [m >> 1][lane][m & 1]
   __shared__ double sR[4][16];
   __shared__ unsigned char sZc[COOP_ZT * 16];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, hl = lane & 15;
   const int sub = blockIdx.x & 3, tile = (blockIdx.x >> 2) % a.n_tiles, iclass = (blockIdx.x >> 2) / a.n_tiles;
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y + 16 * sub;
   const int hend = as_const(a.gene_off)[gene + 1];
   if (h0 >= hend) return;
   const int h = h0 + hl;
   const bool valid = h < hend;
   const int hc = valid ? h : hend - 1;
   const long pset = (long)gene * a.K + iclass;
   const double *Pint = a.pint + pset * a.n_nodes * 4096;
   const long tipstride = a.tip_words;
   const double *Ptip = a.ptip + pset * a.n_nodes * tipstride;
   const int n = a.n;
   if (a.first_matmul >= 0) stage_p<4>(Pint + (long)a.first_matmul * 4096, sP[0], wave, lane);
   const bool zl = a.n_tips <= COOP_ZT;
   if (zl) {
      for (int i = tid; i < a.n_tips * 16; i += 256) sZc[i] = a.z[(long)(i >> 4) * a.z_stride + min(h0 + (i & 15), hend - 1)];
      __syncthreads();
   }
#define TIP_CODE(tip) (zl ? (int)sZc[(tip)*16 + hl] : (int)a.z[(long)(tip)*a.z_stride + hc])
   // this wave's quarter (elements m = 4 wave + r) of the row (code, q) of a tip's column table: pieces 2 wave, 2 wave + 1
#define COOP_TIP(tip, V0, V1)                                                                                     \
   do {                                                                                                         \
      const int row_ = TIP_CODE(tip) * 4 + q, swz_ = TIP_SWZ(row_);                                             \
      const double2 *pt_ = (const double2 *)(Ptip + (long)(tip)*tipstride + row_ * 16);                         \
      V0 = pt_[(2 * wave) ^ swz_]; V1 = pt_[(2 * wave + 1) ^ swz_];                                             \
   } while (0)
   // the rows of the NEXT tip step (tips ta, tb; -1: not requested), asked for when the product in front of it starts
   double2 tv0 = {0, 0}, tv1 = {0, 0}, tw0 = {0, 0}, tw1 = {0, 0};
   int ta = -1, tb = -1;
#define COOP_TIP_AHEAD(FROM)                                                                                      \
   do {                                                                                                         \
      ta = tb = -1;                                                                                             \
      for (int k_ = (FROM); zl; k_++) {                                                                         \
         const Op o_ = fetch_op(a.ops, k_);                                                                     \
         if (o_.code == OP_END || o_.code == OP_MATMUL || o_.code == OP_MATMUL_POP) break;                      \
         const bool two_ = o_.code == OP_SET_TIP2 || o_.code == OP_MUL_TIP2;                                    \
         if (two_ || o_.code == OP_SET_TIP || o_.code == OP_MUL_TIP) {                                          \
            ta = o_.a; COOP_TIP(ta, tv0, tv1);                                                                  \
            if (two_) { tb = o_.b; COOP_TIP(tb, tw0, tw1); }                                                    \
            break;                                                                                              \
         }                                                                                                      \
      }                                                                                                         \
   } while (0)
#define COOP_SLOT(SLOT, STMT)                                                                                     \
   switch (SLOT) {                                                                                              \
   case 0: { double(&S)[4] = st[0]; STMT } break; case 1: { double(&S)[4] = st[1]; STMT } break;                \
   case 2: { double(&S)[4] = st[2]; STMT } break; case 3: { double(&S)[4] = st[3]; STMT } break;                \
   case 4: { double(&S)[4] = st[4]; STMT } break; case 5: { double(&S)[4] = st[5]; STMT } break;                \
   case 6: { double(&S)[4] = st[6]; STMT } break; default: { double(&S)[4] = st[7]; STMT } break;               \
   }
   double cur[4] = {0, 0, 0, 0}, st[COOP_SLOTS][4];
   double lnscale = 0;
   int buf = 0, xb = 0;
   for (int ip = 0;; ip++) {
      const Op op = fetch_op(a.ops, ip);
      if (op.code == OP_END) break;
      switch (op.code) {
      case OP_INIT_ONES: {
#pragma unroll
         for (int r = 0; r < 4; r++) cur[r] = (4 * (4 * wave + r) + q < n) ? 1.0 : 0.0;
      } break;
      case OP_INIT_TIP: {
         const int code = TIP_CODE(op.a);
#pragma unroll
         for (int r = 0; r < 4; r++) cur[r] = (a.cleandata && 4 * (4 * wave + r) + q == code) ? 1.0 : 0.0;
      } break;
      case OP_SET_TIP:
      case OP_MUL_TIP: {
         if (ta != op.a) COOP_TIP(op.a, tv0, tv1);      // (not requested ahead)
         if (op.code == OP_SET_TIP) { cur[0] = tv0.x; cur[1] = tv0.y; cur[2] = tv1.x; cur[3] = tv1.y; }
         else { cur[0] *= tv0.x; cur[1] *= tv0.y; cur[2] *= tv1.x; cur[3] *= tv1.y; }
         ta = tb = -1;
      } break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2: {
         if (ta != op.a) COOP_TIP(op.a, tv0, tv1);
         if (tb != op.b) COOP_TIP(op.b, tw0, tw1);
         if (op.code == OP_SET_TIP2) { cur[0] = tv0.x * tw0.x; cur[1] = tv0.y * tw0.y; cur[2] = tv1.x * tw1.x; cur[3] = tv1.y * tw1.y; }
         else {
            cur[0] = (cur[0] * tv0.x) * tw0.x; cur[1] = (cur[1] * tv0.y) * tw0.y;
            cur[2] = (cur[2] * tv1.x) * tw1.x; cur[3] = (cur[3] * tv1.y) * tw1.y;
         }
         ta = tb = -1;
      } break;
      case OP_PUSH: {
         COOP_SLOT(op.b, { _Pragma("unroll") for (int r = 0; r < 4; r++) S[r] = cur[r]; })
      } break;
      case OP_SCALE: {      // NodeScale treesub.c:7200-7230: the maximum over all the states of the pattern = over the four waves' quarters
         double mx = 0;
#pragma unroll
         for (int r = 0; r < 4; r++) mx = cur[r] > mx ? cur[r] : mx;
         double o = __shfl_xor(mx, 16);
         mx = o > mx ? o : mx;
         o = __shfl_xor(mx, 32);
         mx = o > mx ? o : mx;
         if (q == 0) sR[wave][hl] = mx;
         __syncthreads();
#pragma unroll
         for (int w2 = 0; w2 < 4; w2++) { const double v = sR[w2][hl]; mx = v > mx ? v : mx; }
         __syncthreads();
         double fac;
         if (mx < 1e-300) {
#pragma unroll
            for (int r = 0; r < 4; r++) cur[r] = (4 * (4 * wave + r) + q < n) ? 1.0 : 0.0;
            fac = -800;
         }
         else {
#pragma unroll
            for (int r = 0; r < 4; r++) cur[r] /= mx;
            fac = log(mx);
         }
         lnscale += fac;
      } break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         double2 *xs = (double2 *)sX[xb];
         xs[(2 * wave) * 64 + lane] = make_double2(cur[0], cur[1]);
         xs[(2 * wave + 1) * 64 + lane] = make_double2(cur[2], cur[3]);
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         __syncthreads();      // the operand partial is complete, this branch's P has landed in sP[buf], every wave is done with sP[buf ^ 1]
         COOP_TIP_AHEAD(ip + 1);      // the tip step behind this product, if there is one: its rows arrive under the MFMAs
         if (op.c >= 0) stage_p<4>(Pint + (long)op.c * 4096, sP[buf ^ 1], wave, lane);
         const double2 *xr = (const double2 *)sX[xb], *sp = (const double2 *)sP[buf];
         double2 xv[8], af[8];
#pragma unroll
         for (int p = 0; p < 8; p++) { xv[p] = xr[p * 64 + lane]; af[p] = sp[(p * 4 + wave) * 64 + lane]; }
         v4d acc = {0, 0, 0, 0};
#pragma unroll
         for (int kb2 = 0; kb2 < 8; kb2++) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kb2].x, xv[kb2].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kb2].y, xv[kb2].y, acc, 0, 0, 0);
         }
         const int pop = mm_pop_slot(op), push = mm_push_slot(op);
         double y[4] = {acc[0], acc[1], acc[2], acc[3]};
         if (pop >= 0) { COOP_SLOT(pop, { _Pragma("unroll") for (int r = 0; r < 4; r++) y[r] = S[r] * y[r]; }) }
         if (push >= 0) { COOP_SLOT(push, { _Pragma("unroll") for (int r = 0; r < 4; r++) S[r] = y[r]; }) }
         else {
#pragma unroll
            for (int r = 0; r < 4; r++) cur[r] = y[r];
         }
         buf ^= 1; xb ^= 1;
      } break;
      case OP_ROOT: {      // the whole partial to wave 0, which sums it as the other kernels do (MFMA_ROOT_CASE)
         double2 *xs = (double2 *)sX[xb];
         xs[(2 * wave) * 64 + lane] = make_double2(cur[0], cur[1]);
         xs[(2 * wave + 1) * 64 + lane] = make_double2(cur[2], cur[3]);
         __syncthreads();
         if (wave == 0) {
            const double2 *xr = (const double2 *)sX[xb];
            const double *pq = a.pi + (long)(a.n_pi > 1 ? gene : 0) * 64 + q * 16;
            double f = 0;
#pragma unroll
            for (int p = 0; p < 8; p++) { const double2 v = xr[p * 64 + lane]; f = fma(pq[2 * p], v.x, f); f = fma(pq[2 * p + 1], v.y, f); }
            f += __shfl_xor(f, 16);
            f += __shfl_xor(f, 32);
            if (q == 0 && valid) {
               double out = 0;
               if (a.weights[h] > 0) out = root_value(a, f, lnscale);
               a.fhK[(long)iclass * a.n_patt + h] = out;
            }
         }
         xb ^= 1;
      } break;
      default: break;
      }
   }
#undef TIP_CODE
#undef COOP_TIP
#undef COOP_TIP_AHEAD
#undef COOP_SLOT
}

// ---- mfma64 "stream": the production kernel (<= MFMA_ZT tips, <= 64 character codes, register stack).
// Every operand the tree walk consumes — the P of an internal branch in MFMA order, or the whole column
// table of a tip branch — is one 32 KB block, and the program fixes the order in which blocks are used.
// 8 waves (128 patterns) per workgroup share a ring of four 32 KB LDS buffers that a linear LDS-DMA
// stream keeps filled three blocks ahead of use (4 x buffer_load_dwordx4 ... lds per wave per block),
// so neither P nor tip data is ever waited for at L2 latency, no VGPRs hold data in flight, and every
// DMA instruction is a fully coalesced 1 KB line burst.  Tip factors are then LDS gathers (rows are
// XOR-swizzled by pmat_kernel so random rows spread over the banks); one s_barrier per step.
__global__ __launch_bounds__(512, 2) void prune_mfma64_stream(PruneArgs a)
{
   constexpr int WAVES = 8, TP = 128;
   extern __shared__ __attribute__((aligned(16))) unsigned char smem_stream[];
   double *ring = (double *)smem_stream;                        // [4][4096]
   unsigned char *sZ = (unsigned char *)(ring + 4 * 4096);      // [n_tips][128]
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int hl = lane & 15;
   const int tile = blockIdx.x % a.n_tiles, iclass = blockIdx.x / a.n_tiles;
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y;
   const int hend = as_const(a.gene_off)[gene + 1];
   const int hw = wave * 16 + hl;          // pattern within the tile
   const int h = h0 + hw;
   const bool valid = h < hend;
   const long pset = (long)gene * a.K + iclass;
   const double *Pint = a.pint + pset * a.n_nodes * 4096;
   const long tipstride = 4096;            // one 32 KB block per tip (n_codes <= 64)
   const double *Ptip = a.ptip + pset * a.n_nodes * tipstride;
   const int n = a.n;
   const StreamBlk *stream = (const StreamBlk *)a.stream;
   const int nblk = a.n_stream;

   PROF_STAMP(a.prof_stride - 1);
   int issued = 0, consumed = 0;
#define STREAM_ISSUE()                                                                                           \
   do {                                                                                                         \
      const long long sb = ((const CONST_AS long long *)(unsigned long long)stream)[issued];                    \
      const int is_tip = (int)(sb & 0xffffffff), node = (int)(sb >> 32);                                        \
      const double *src = is_tip ? Ptip + (long)node * tipstride : Pint + (long)node * 4096;                    \
      stage_p<WAVES>(src, ring + (issued & 3) * 4096, wave, lane);                                              \
      issued++;                                                                                                 \
   } while (0)
   for (int i = 0; i < 3; i++)
      if (issued < nblk) STREAM_ISSUE();
   {
      const int nz = a.n_tips * TP;
      for (int idx = tid; idx < nz; idx += WAVES * 64) {
         const int tip = idx / TP, hh = idx % TP;
         const int hx = h0 + hh < hend ? h0 + hh : hend - 1;
         sZ[idx] = a.z[(long)tip * a.z_stride + hx];
      }
   }
   __syncthreads();   // publish sZ (INIT_TIP may read it before the first stream step)

   double cur[16], s0[16], s1[16];   // every program writes cur/s0/s1 (INIT/SET/PUSH) before reading them
   double lnscale = 0;
#define TIP_CODE(tip) ((int)sZ[(tip)*TP + hw])
   // consume the next c blocks of the stream: they have landed for every wave after this returns, and the
   // buffers used by the previous step are refilled with the blocks 3..4 ahead
#ifdef ABL_NO_BARRIER
#define STREAM_BARRIER()
#else
#define STREAM_BARRIER() __syncthreads()
#endif
#define STREAM_STEP(c)                                                                                           \
   do {                                                                                                         \
      wait_blocks_in_flight(issued - (consumed + (c)));                                                         \
      STREAM_BARRIER();                                                                                         \
      while (issued < consumed + 4 && issued < nblk) STREAM_ISSUE();                                            \
   } while (0)

   PROF_STAMP(0);
   const int lane0 = lane;
   for (int ip = 0;; ip++) {
      const Op op = fetch_op(a.ops, ip);
      PROF_STAMP(1 + ip);
      if (op.code == OP_END) break;
      // re-derive every per-lane address from an opaque copy of the lane id inside the loop: cheap VALU,
      // and nothing loop-invariant is left for LICM to hoist into (spilled) VGPRs
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      const int q = lane >> 4;
      switch (op.code) {
         MFMA_CORE_CASES()
         MFMA_ROOT_CASE()
      case OP_MUL_TIP:
      case OP_SET_TIP: {
         double2 v[8];
         STREAM_STEP(1);
         tip_lds(ring + (consumed & 3) * 4096, TIP_CODE(op.a), q, lane, v);
         consumed += 1;
         if (op.code == OP_SET_TIP) {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] = v[i].x; cur[2 * i + 1] = v[i].y; }
         }
         else {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] *= v[i].x; cur[2 * i + 1] *= v[i].y; }
         }
      } break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2: {
         double2 v[8], w[8];
         STREAM_STEP(2);
         tip_lds(ring + (consumed & 3) * 4096, TIP_CODE(op.a), q, lane, v);
         tip_lds(ring + ((consumed + 1) & 3) * 4096, TIP_CODE(op.b), q, lane, w);
         consumed += 2;
         if (op.code == OP_SET_TIP2) {
#pragma unroll
            for (int i = 0; i < 8; i++) { cur[2 * i] = v[i].x * w[i].x; cur[2 * i + 1] = v[i].y * w[i].y; }
         }
         else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
               cur[2 * i] = (cur[2 * i] * v[i].x) * w[i].x;
               cur[2 * i + 1] = (cur[2 * i + 1] * v[i].y) * w[i].y;
            }
         }
      } break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         STREAM_STEP(1);
         PROF_SUB(1, ip);
         v4d acc[4];
         mfma_matvec(ring + (consumed & 3) * 4096, lane, cur, acc);
         consumed += 1;
         PROF_SUB(2, ip);
         MFMA_EPILOGUE_REG();
      } break;
      default: break;
      }
   }
#undef TIP_CODE
#undef STREAM_STEP
#undef STREAM_ISSUE
}

// ---- valu<N>: 4 / 5 / 20 states, one pattern per lane ------------------------------------------
// The partial lives in N registers; P(t) entries are wave-uniform, so the compiler fetches them with
// scalar loads (s_load) and feeds v_fma_f64 from SGPRs: no LDS, no barriers.  The whole tree is walked
// per lane, so only tips (1 B) and the result (8 B) touch HBM unless keep-partials is on.
// REGSTK: the partial stack is addressed through unrolled wave-uniform compares, so it stays in registers; the plain
// form indexes stk[op.b] dynamically, which the compiler can only do through scratch memory (measured on the 20-state
// kernel: 680 MB of scratch writes per launch at 1e5 patterns).  The engine picks the shallowest instantiation that
// fits the tree's stack depth.
template <int N, int MAXD, bool REGSTK = false>
__global__ __launch_bounds__(256) void prune_valu(PruneArgs a)
{
   const int tid = threadIdx.x;
   const int tile = blockIdx.x % a.n_tiles, iclass = blockIdx.x / a.n_tiles;
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y;
   const int hend = as_const(a.gene_off)[gene + 1];
   const int h = h0 + tid;
   const bool valid = h < hend;
   const int hc = valid ? h : hend - 1;
   const long pset = (long)gene * a.K + iclass;
   const double *Pint = a.pint + pset * a.n_nodes * (N * N);
   const long tipstride = a.tip_words;
   const double *Ptip = a.ptip + pset * a.n_nodes * tipstride;

   double cur[N];
   double stk[MAXD][N];
   double lnscale = 0;
#pragma unroll
   for (int j = 0; j < N; j++) cur[j] = 0;

   for (int ip = 0;; ip++) {
      const Op op = fetch_op(a.ops, ip);
      if (op.code == OP_END) break;
      switch (op.code) {
      case OP_INIT_ONES: {
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] = 1.0;
      } break;
      case OP_INIT_TIP: {
         const int code = a.z[(long)op.a * a.z_stride + hc];
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] = (a.cleandata && j == code) ? 1.0 : 0.0;
      } break;
      case OP_MUL_TIP: {
         const int code = a.z[(long)op.a * a.z_stride + hc];
         const double *pt = Ptip + (long)op.a * tipstride + code * N;
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] *= pt[j];
      } break;
      case OP_SET_TIP: {
         const int code = a.z[(long)op.a * a.z_stride + hc];
         const double *pt = Ptip + (long)op.a * tipstride + code * N;
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] = pt[j];
      } break;
      case OP_SET_TIP2:
      case OP_MUL_TIP2: {
         const int c1 = a.z[(long)op.a * a.z_stride + hc], c2 = a.z[(long)op.b * a.z_stride + hc];
         const double *p1 = Ptip + (long)op.a * tipstride + c1 * N;
         const double *p2 = Ptip + (long)op.b * tipstride + c2 * N;
         if (op.code == OP_SET_TIP2) {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] = p1[j] * p2[j];
         }
         else {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] = (cur[j] * p1[j]) * p2[j];
         }
      } break;
      case OP_PUSH: {
         if constexpr (REGSTK) {
#pragma unroll
            for (int d = 0; d < MAXD; d++)
               if (op.b == d) {
#pragma unroll
                  for (int j = 0; j < N; j++) stk[d][j] = cur[j];
               }
         }
         else {
#pragma unroll
            for (int j = 0; j < N; j++) stk[op.b][j] = cur[j];
         }
      } break;
      case OP_MATMUL:
      case OP_MATMUL_POP: {
         const CONST_AS double *P = as_const(Pint + (long)op.a * (N * N));
         double out[N];
#pragma unroll
         for (int j = 0; j < N; j++) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < N; k++) t = fma(P[j * N + k], cur[k], t);
            out[j] = t;
         }
         const int pop = mm_pop_slot(op), push = mm_push_slot(op);
         if constexpr (REGSTK) {
#pragma unroll
            for (int d = 0; d < MAXD; d++)
               if (pop == d) {
#pragma unroll
                  for (int j = 0; j < N; j++) out[j] = stk[d][j] * out[j];
               }
#pragma unroll
            for (int d = 0; d < MAXD; d++)
               if (push == d) {
#pragma unroll
                  for (int j = 0; j < N; j++) stk[d][j] = out[j];
               }
         }
         else {
            if (pop >= 0) {
#pragma unroll
               for (int j = 0; j < N; j++) out[j] = stk[pop][j] * out[j];
            }
            if (push >= 0) {
#pragma unroll
               for (int j = 0; j < N; j++) stk[push][j] = out[j];
            }
         }
         if (push < 0) {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] = out[j];
         }
      } break;
      case OP_SCALE: {
         double mx = 0;
#pragma unroll
         for (int j = 0; j < N; j++) mx = cur[j] > mx ? cur[j] : mx;
         double fac;
         if (mx < 1e-300) {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] = 1.0;
            fac = -800;
         }
         else {
#pragma unroll
            for (int j = 0; j < N; j++) cur[j] /= mx;
            fac = log(mx);
         }
         lnscale += fac;
         if (a.keep && valid) a.scalef[((long)iclass * a.n_scale + op.b) * a.n_patt + h] = fac;
      } break;
      case OP_STORE: {
         if (valid) {
            double *dst = a.partials + (((long)iclass * a.n_int + (op.a - a.n_tips)) * a.n_patt + h) * N;
#pragma unroll
            for (int j = 0; j < N; j++) dst[j] = cur[j];
         }
      } break;
      case OP_LOAD: {
         const double *src = a.partials + (((long)iclass * a.n_int + (op.a - a.n_tips)) * a.n_patt + hc) * N;
#pragma unroll
         for (int j = 0; j < N; j++) cur[j] = src[j];
      } break;
      case OP_EXPORT: {
         if (valid) {
            double *dst = a.export_buf + ((long)iclass * a.n_patt + h) * N;
#pragma unroll
            for (int j = 0; j < N; j++) dst[j] = cur[j];
            if (a.export_scale) a.export_scale[(long)iclass * a.n_patt + h] = lnscale;
         }
      } break;
      case OP_ROOT: {
         const double *pi = a.pi + (long)(a.n_pi > 1 ? gene : 0) * N;
         double f = 0;
#pragma unroll
         for (int j = 0; j < N; j++) f = fma(pi[j], cur[j], f);
         if (a.keep && a.n_scale) {
            lnscale = 0;
            if (valid)
               for (int k = 0; k < a.n_scale; k++) lnscale += a.scalef[((long)iclass * a.n_scale + k) * a.n_patt + h];
         }
         if (valid) {
            double out = 0;
            if (a.weights[h] > 0) out = root_value(a, f, lnscale);
            a.fhK[(long)iclass * a.n_patt + h] = out;
         }
      } break;
      default: break;
      }
   }
}

// Tile blocks of the specialised kernel (PruneArgs::ztiles): per 128-pattern tile the tip codes of its patterns, one
// 128-byte row per tip, then a row of weight > 0 flags; patterns past the tile's gene read as code 0 / flag 0.
// tip_of / row_at (trees of more than 207 tips, jit.h: jit_zplan): the rows in the order the walk consumes them — row r holds tip
// tip_of[r] (n_tips: the flags) and sits at row position row_at[r] of the tile's block (the block is cut into pieces of equal size, a
// piece's rows from its start).  null: rows in tip order, one block.
__global__ __launch_bounds__(256) void ztile_kernel(const int2 *tiles, const int *gene_off, const unsigned char *z, long z_stride,
                                                    const double *weights, int n_tips, int zt_bytes, unsigned char *out, const int *tip_of, const int *row_at)
{
   const int t = blockIdx.x, i = threadIdx.x, tp = blockDim.x;      // one thread per pattern of the tile (128 or 192)
   const int g = tiles[t].x, h = tiles[t].y + i, hend = gene_off[g + 1];
   unsigned char *o = out + (long)t * zt_bytes;
   for (int k = i; k < zt_bytes; k += tp) o[k] = 0;
   __syncthreads();
   for (int r = 0; r <= n_tips; r++) {
      const int tip = tip_of ? tip_of[r] : r;
      const long at = (long)(tip_of ? row_at[r] : r) * tp;
      unsigned char v = 0;
      if (h < hend) v = tip < n_tips ? z[tip * z_stride + h] : (unsigned char)(weights[h] > 0 ? 1 : 0);
      o[at + i] = v;
   }
}

// Tip codes pattern-major for the fused one-pattern-per-lane kernels: row h = zw dwords, byte t = code of tip t.
__global__ __launch_bounds__(256) void zpm_kernel(const unsigned char *z, long z_stride, int n_tips, int n_patt, int zw, unsigned int *out)
{
   const long h = (long)blockIdx.x * 256 + threadIdx.x;
   if (h >= n_patt) return;
   for (int w = 0; w < zw; w++) {
      unsigned int v = 0;
      for (int b = 0; b < 4; b++) {
         const int t = 4 * w + b;
         if (t < n_tips) v |= (unsigned int)z[t * z_stride + h] << (8 * b);
      }
      out[h * zw + w] = v;
   }
}

}  // namespace paml_amd
