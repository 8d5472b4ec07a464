// device_common.h — device-side definitions shared by the ahead-of-time kernels (kernels_*.h) and the
// per-tree specialised kernels that jit.h generates and compiles with hiprtc.  No host/STL headers here.
#pragma once
#ifndef __HIPCC_RTC__   // hiprtc pre-includes the HIP device runtime
#include <hip/hip_runtime.h>
#endif

#include "../../include/paml_amd.h"

namespace paml_amd {

typedef double v4d __attribute__((ext_vector_type(4)));

enum OpCode : int {
   OP_INIT_ONES = 0, OP_INIT_TIP = 1, OP_MUL_TIP = 2, OP_PUSH = 3, OP_MATMUL = 4, OP_MATMUL_POP = 5,
   OP_SCALE = 6, OP_STORE = 7, OP_LOAD = 8, OP_ROOT = 9, OP_END = 10,
   // fused forms produced by the peephole pass (same arithmetic, fewer dependent memory round trips):
   OP_SET_TIP = 11,    // cur = tipcol(a)                 == INIT_ONES ; MUL_TIP a
   OP_SET_TIP2 = 12,   // cur = tipcol(a) * tipcol(b)     == INIT_ONES ; MUL_TIP a ; MUL_TIP b   (a cherry)
   OP_MUL_TIP2 = 13,   // cur *= tipcol(a) * tipcol(b)    == MUL_TIP a ; MUL_TIP b
   OP_EXPORT = 14      // write cur as [class][pattern][state] to PruneArgs::export_buf (branch-local evaluation)
};

struct Op { int code, a, b, c; };   // a: node/tip, b: stack slot / scale slot / 2nd tip, c: prefetch link (-1 none)
// MATMUL / MATMUL_POP encode two stack slots in b: bits 0..7 = (slot popped + 1), bits 8..15 = (slot the
// result is pushed to + 1); 0 = none.  A push slot means "MATMUL[_POP] ; PUSH" fused: the result goes
// straight to the stack slot and `cur` is dead until the next INIT/SET op.
// Prefetch links: for MATMUL ops c = son of the next MATMUL (its P is staged while this one computes);
// for tip ops c = the next tip in program order (its column table is fetched ahead of use).
__host__ __device__ inline int mm_pop_slot(const Op &o) { return (o.b & 0xff) - 1; }
__host__ __device__ inline int mm_push_slot(const Op &o) { return ((o.b >> 8) & 0xff) - 1; }


struct PruneArgs {
   const Op *ops;
   const unsigned char *z;     // [n_tips][z_stride]
   long z_stride;
   const int2 *tiles;          // (gene, first pattern) per tile
   int n_tiles;
   const int *gene_off;
   const double *weights;
   int n, n_tips, n_nodes, K, n_genes, n_codes, cleandata, n_pi, mode, n_scale, keep, n_patt;
   const double *pi;           // VALU: [n_pi][n];  mfma64: [n_pi][4][16] (q-major, zero padded)
   const double *pint;         // per (pset, node): n*n row-major (VALU) or 4096 frag (mfma64)
   const double *ptip;         // per (pset, node): n_codes * tipw
   double *fhK;                // [K][n_patt]
   double *partials;           // keep mode
   double *scalef;             // keep mode: [K][n_scale][n_patt]
   double *stack_scratch;      // overflow stack (mfma64)
   int stack_overflow_slots;
   int first_matmul;
   int n_int;                  // n_nodes - n_tips
   int first_tip;              // first tip whose column table is consumed
   const void *stream;         // stream kernel: operand blocks in order of use, {is_tip, node} pairs
   int n_stream;
   long tip_words;             // doubles per tip table
   double *export_buf;         // OP_EXPORT target: [K][n_patt][n]
   double *export_scale;       // OP_EXPORT: summed scale factors of the exported partial [K][n_patt] (null: none)
   unsigned long long *prof;   // PROF_OPS builds only: [block][op] s_memtime stamps of thread 0
   int prof_stride, prof_tid;
   const double *pcol;         // jit kernel, 61 states: column 60 of every P, [pset][n_nodes][q][m] (rank-1 tail of the matmul)
   double *fscale;             // jit kernel with scaling nodes: summed scale factors [K][n_patt] (the log is taken later)
   const unsigned char *ztiles; // jit kernel: per tile, (n_tips + 1) rows of 128 bytes (tip codes of the tile's patterns,
   int zt_bytes;                // then the weight > 0 flags), zero padded to zt_bytes (a multiple of 2048)
   // ---- kernels with the reduction fused in (one workgroup = one reduction chunk of patterns; jit_generate_valu_fused) ----
   const unsigned int *zpm;     // tip codes pattern-major: [n_patt][zpm_words] dwords, 4 codes per dword (tip t = byte t)
   int zpm_words;               // dwords per pattern (a multiple of 4)
   int Km;                      // classes of the model (K = Km x batch elements; blockIdx.y = batch element)
   int chunk;                   // patterns per workgroup = per partial sum
   int first_chunk, nb_stride;  // partial sum of (batch b, chunk c) lives at red_partial[b * nb_stride + first_chunk + c]
   int want_fhk;                // store fhK (always stored with scaling nodes: the class mixture needs the logs)
   const double *freqK;         // [Km] (+ b * freqK_bs)
   long freqK_bs;
   double *lnf;                 // optional [batch][n_patt]
   double *red_partial;
   double *red_out;             // [batch]
   int *red_counter;            // [batch] zeroed; non-null: the last workgroup to finish adds the partials up (one GPU)
   int nb_local;                // chunks of this engine: a workgroup walks the chunks blockIdx.x, blockIdx.x + gridDim.x, ... (batched
                                // evaluations run fewer, longer workgroups per element: the LDS tables are filled once per workgroup)
   // resident partials of the 21..64-state kernels (STORE / LOAD): ONE layout whatever kernel writes or reads them — per class and internal
   // node `part_groups` 16-pattern groups in the order of the 64-pattern tile table (group = 4 x tile + wave there); a kernel with larger
   // tiles finds the group of its tile's first 16 patterns in tile_group0[tile] (a 128-pattern tile is two consecutive 64-pattern tiles of its gene)
   const int *tile_group0;
   int part_groups;
   double *part_dump;           // 8 x 1024 doubles nobody reads: where the per-tree kernel's waves whose 16 patterns lie past their gene's end
                                // put their STOREs (every wave must issue the same vector-memory operations: the operand ring's waits count them)
   const unsigned long long *code_mask;      // [n_codes] state sets of the character codes as bit masks (tools.c:20 nChara / CharaMap): the per-tree
                                             // kernel's codes beyond the 64 a ring block has rows for (JIT_AMB_OVERFLOW)
};

__device__ __forceinline__ double root_value(const PruneArgs &a, double f, double lnscale)
{
   // fx_r treesub.c:7731-7749 / lfun 7782-7798: floors then log + scale factors
   if (f <= 0) f = (a.mode == PAML_AMD_MODE_LFUN ? 1e-80 : 1e-300);
   if (a.mode == PAML_AMD_MODE_LFUN || a.n_scale) f = log(f) + lnscale;
   return f;
}


// Resident partials of the 21..64-state kernels (keep-partials STORE / LOAD, the branch-local evaluation, its coefficients): per
// 16-pattern group 1024 doubles, element (m, lane) = state 4m + (lane >> 4) of pattern lane & 15 at ((m >> 1) * 64 + lane) * 2 + (m & 1)
// — a lane's two consecutive m are one 16-byte access, a wave instruction moves 1 KB (eight per partial; 8-byte accesses, sixteen
// per partial, are issue-bound well below the HBM rate: MI355X_MICROARCH.md, store tail).
typedef double part2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void part_load(const double *p, int lane, double (&x)[16])
{
   const part2_t *p2 = (const part2_t *)p + lane;
#pragma unroll
   for (int i = 0; i < 8; i++) { const part2_t v = p2[i * 64]; x[2 * i] = v.x; x[2 * i + 1] = v.y; }
}
__device__ __forceinline__ void part_store(double *p, int lane, const double (&x)[16])
{
   part2_t *p2 = (part2_t *)p + lane;
#pragma unroll
   for (int i = 0; i < 8; i++) p2[i * 64] = (part2_t){x[2 * i], x[2 * i + 1]};
}
__host__ __device__ inline int part_index(int m, int lane) { return (((m >> 1) * 64 + lane) << 1) | (m & 1); }

#define MFMA_RS 2      // register stack slots; deeper slots spill to global scratch
#define MFMA_ZT 128    // most tips whose codes the dma kernel keeps in LDS

// Wave-uniform, read-only data (the tree program, tile table, P(t) entries of the VALU kernels) is read
// through the constant address space so the compiler uses scalar loads (s_load, lgkmcnt) instead of a
// vector load + vmcnt(0) wait that would also drain the in-flight LDS-DMA prefetches.
#define CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const CONST_AS T *as_const(const T *p)
{
   return (const CONST_AS T *)(unsigned long long)p;
}
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Op fetch_op(const Op *ops, int ip)
{
   const v4i o = ((const CONST_AS v4i *)(unsigned long long)ops)[ip];   // one s_load_dwordx4
   return Op{o.x, o.y, o.z, o.w};
}

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// LDS-DMA through a raw buffer descriptor: address = SGPR descriptor base + 32-bit VGPR offset + SGPR
// offset, so no 64-bit per-lane pointers exist for the compiler to hoist and spill.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
   return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes, 0x00020000);
}

// One LDS-DMA instruction: 64 lanes x 16 B from (descriptor base + voff + soff) to the wave-uniform LDS
// address `lds` (+ lane*16).  Issued from inline asm on purpose: hipcc treats a visible LDS-DMA as a pending
// LDS write and puts s_waitcnt vmcnt(0) in front of EVERY later ds_read and barrier, which would serialise
// the prefetch stream with its consumers.  All ordering is explicit instead: counted s_waitcnt vmcnt(N)
// (loads retire in issue order) followed by a workgroup barrier before any wave reads the landed block.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, const void *lds, int voff, int soff)
{
   const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)lds;
   asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                :
                : "s"(la), "v"(voff), "s"(r), "s"(soff)
                : "memory", "m0");
}

// Same with one dword per lane (256 B per wave instruction): small blocks such as a tile's tip codes.
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t r, const void *lds, int voff, int soff)
{
   const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)lds;
   asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                :
                : "s"(la), "v"(voff), "s"(r), "s"(soff)
                : "memory", "m0");
}

// Stage one 32 KB block (a P in MFMA operand order, or a tip's column table) global -> LDS: each wave
// instruction moves 1 KiB to a wave-uniform LDS base, no VGPR round trip.
template <int WAVES>
__device__ __forceinline__ void stage_p(const double *g, double *s, int wave, int lane)
{
#ifdef ABL_NO_STAGE
   return;
#endif
   const __amdgpu_buffer_rsrc_t r = make_rsrc(g, 32768);
#pragma unroll
   for (int c = 0; c < 32 / WAVES; c++) {
      const int chunk = c * WAVES + wave;   // wave-uniform
      dma16(r, (const char *)s + chunk * 1024, lane * 16, chunk * 1024);
   }
}

// 64 MFMAs: acc = P . cur, with P's fragments read from LDS one k-block pair ahead of their use.
__device__ __forceinline__ void mfma_matvec(const double *sPbuf, int lane, const double (&cur)[16], v4d (&acc)[4])
{
   const double2 *sp = (const double2 *)sPbuf;
#pragma unroll
   for (int jb = 0; jb < 4; jb++) acc[jb] = (v4d){0, 0, 0, 0};
#ifdef ABL_NO_MFMA
#pragma unroll
   for (int jb = 0; jb < 4; jb++) {
      const double2 a2 = sp[jb * 64 + lane];
      acc[jb] = (v4d){a2.x * cur[4 * jb], a2.y * cur[4 * jb + 1], a2.x * cur[4 * jb + 2], a2.y * cur[4 * jb + 3]};
   }
#else
   double2 af[2][4];
#pragma unroll
   for (int jb = 0; jb < 4; jb++) af[0][jb] = sp[jb * 64 + lane];
#pragma unroll
   for (int kb2 = 0; kb2 < 8; kb2++) {
      if (kb2 + 1 < 8) {
#pragma unroll
         for (int jb = 0; jb < 4; jb++) af[(kb2 + 1) & 1][jb] = sp[((kb2 + 1) * 4 + jb) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the next pair's ds_reads ahead of this pair's MFMAs
#pragma unroll
      for (int jb = 0; jb < 4; jb++)
         acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kb2 & 1][jb].x, cur[2 * kb2], acc[jb], 0, 0, 0);
#pragma unroll
      for (int jb = 0; jb < 4; jb++)
         acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kb2 & 1][jb].y, cur[2 * kb2 + 1], acc[jb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
   }
#endif
}


// XOR swizzle of the eight 16-byte pieces of a tip-table row (row = code * 4 + state quarter): lanes of one
// ds_read_b128 group read rows of 16 different patterns, i.e. random codes; mixing the code's low and middle bits into
// the slot spreads them over all 8 slots of their bank half instead of 4.
// (TIP_SWZ_OFF: build-time experiment — plain rows, the eight pieces at immediate offsets of one address register: eight fewer
//  vector instructions per gathered row against more bank conflicts.  Measured slower, 1.577 against 1.555 ms per launch at 16 taxa x
//  10^6 codon patterns (profiles/r04_61state.txt); PAML_AMD_EXTRA_FLAGS=-DTIP_SWZ_OFF=1, see engine.py build())
#ifdef TIP_SWZ_OFF
#define TIP_SWZ(row) 0
#else
#define TIP_SWZ(row) ((((row) >> 2) ^ ((row) >> 5)) & 7)
#endif
struct StreamBlk { int is_tip, node; };

__device__ __forceinline__ void wait_blocks_in_flight(int n)   // allow the n newest blocks (4 loads each) to fly
{
   if (n >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
   else if (n == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
   else if (n == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
   else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// a tip's factors gathered straight from its (L2-resident) column table in global memory
__device__ __forceinline__ void tip_gather(const double *Ptip, long tipstride, int tip, int code, int q, double2 (&v)[8])
{
   const int row = code * 4 + q, swz = TIP_SWZ(row);
   const double2 *pt = (const double2 *)(Ptip + (long)tip * tipstride + row * 16);
#pragma unroll
   for (int i = 0; i < 8; i++) v[i] = pt[i ^ swz];     // piece i lives in slot i ^ swz (see pmat_kernel)
}

__device__ __forceinline__ void tip_lds(const double *tab, int code, int q, int lane, double2 (&v)[8])
{
#ifdef ABL_NO_TIPLOAD
#pragma unroll
   for (int p = 0; p < 8; p++) v[p] = make_double2(0.5 + code * 1e-3, 0.25 + q * 1e-3);
   return;
#endif
   const int row = code * 4 + q, swz = TIP_SWZ(row);
   const char *base = (const char *)tab + row * 128;
#pragma unroll
   for (int p = 0; p < 8; p++) v[p] = *(const double2 *)(base + ((p ^ swz) * 16));
}


// ---- more than 64 character codes (JIT_AMB_OVERFLOW: defined by the generator when the data set has them) ----------------------------
// A ring block holds a tip's rows of the codes 0 .. 63 (32 KB).  61 sense codons leave three of them to ambiguous triplets; a data set
// with more (SetMapAmbiguity treesub.c:1218-1286: every distinct ambiguous triplet of a cleandata = 0 alignment is a code of its own) keeps
// the others out of the block — paml_amd_set_tips numbers the ambiguous codes by frequency x set size, so the ones left out are the rare
// and the small — and a lane that meets one adds up the rows of the code's states itself: the states are single-state codes < 64, the sum
// runs over them in ascending order from 0, the order in which pmat_mfma_kernel forms a table row from CharaMap (codeml.c:3560-3567;
// the engine takes this path only when every such code lists its states in ascending order, as SetMapAmbiguity does) — the same bits.
// The state set is a 64-bit mask: from LDS (the unused three quarters of sPi, when there is one frequency vector) or one global load;
// then as many LDS row reads as the set has states, inside a divergent branch that most waves skip.  (A global load here is younger
// than every DMA piece in flight, so the counted waits of the operand ring stay valid.)
struct JitAmb {
   const unsigned long long *mask_g;      // [n_codes] in global memory
   __attribute__((address_space(3))) const unsigned long long *mask_s;      // codes 64 .. in LDS (index code - 64), read when `lds`
   bool lds;      // (typed as an LDS pointer: through a generic one the compiler merges the two reads into ONE flat load of a selected address,
                  //  and a flat load's wait drains the ring's DMA stream — measured: + 11 % per evaluation for 0.45 % of the cells)
};
#ifdef JIT_AMB_OVERFLOW
#define JIT_FASTCODE(C) ((C) & 63)
// (inlined at every tip step: as a function of its own — tried — the fast path is no faster and the rare path pays the call: 1.89 against 1.73 ms)
template <int NP>
__device__ __forceinline__ void jit_tip_overflow(const JitAmb &amb, const double *tab, int code, int q, double2 (&v)[8])
{
#pragma unroll
   for (int p = 0; p < 8; p++) v[p] = make_double2(0.0, 0.0);
   unsigned long long m;
   if (amb.lds) m = amb.mask_s[code - 64];
   else m = amb.mask_g[code];
   // two states per turn, their rows requested together (half the LDS round trips of one state after the other; four per turn spill);
   // a turn's missing state adds + 0.0, which changes nothing, and the additions keep the ascending order
   while (m) {
      int st[2];
      bool on[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
         on[u] = m != 0;
         st[u] = on[u] ? __builtin_ctzll(m) : 0;
         m = on[u] ? (m & (m - 1)) : 0;
      }
      double2 t[2][NP];
#pragma unroll
      for (int u = 0; u < 2; u++) {
         const int row = st[u] * 4 + q, swz = TIP_SWZ(row);
         const char *base = (const char *)tab + row * 128;
#pragma unroll
         for (int p = 0; p < NP; p++) t[u][p] = *(const double2 *)(base + ((p ^ swz) * 16));
      }
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
         for (int p = 0; p < NP; p++) {
            v[p].x += on[u] ? t[u][p].x : 0.0;
            v[p].y += on[u] ? t[u][p].y : 0.0;
         }
   }
}
#else
#define JIT_FASTCODE(C) (C)
#endif

// ------------------------------------------------------------------------------------------------
// Building blocks of the per-tree specialised kernel (jit.h emits a straight-line sequence of these).
// A partial is a v4d[4]: element m (state 4m + q of this lane's pattern) is x[m >> 2][m & 3] — exactly
// the accumulator tuple of row block jb = m >> 2, so MFMA results are partials with no copies, and the
// B operand of k-block kb is x[kb >> 2][kb & 3].
// ------------------------------------------------------------------------------------------------
/* Waves per workgroup of the per-tree kernel = 16-pattern groups per tile.  8 (two per SIMD) or 12 (three per SIMD, <= 168
 * VGPRs): with three, a SIMD's matrix pipe finds a wave with an MFMA ready more often, and the per-step synchronisation is
 * spread over half as much again of arithmetic. */
#ifndef JIT_WAVES
#define JIT_WAVES 8
#endif
#define JIT_TP (JIT_WAVES * 16)
#ifdef JIT_ABL_NOBAR
#define JIT_SYNC() ((void)0)
#else
#define JIT_SYNC() __syncthreads()
#endif
// `side(kb2)` runs once per k-block pair between the two MFMA groups: the generator puts the ring's refill DMAs there
// (a few per iteration) so that their issue — which can queue behind the other waves' — never delays the first MFMAs.
struct JitNoSide { __device__ __forceinline__ void operator()(int) const {} };
// 61 states: the sixteenth k-block of P holds the single column 60.  With `col` (that column as col[q][m] = P[4m+q][60])
// its contribution y[m] += P[4m+q][60] * x[60] seeds the accumulators as sixteen v_mul_f64 and the four MFMAs of that
// k-block are skipped (x[60] lives in element 15 of the q = 0 lane of the pattern: one cross-lane read).
__device__ __forceinline__ double jit_x60(const v4d (&x)[4], int lane) { return __shfl(x[3][3], lane & 15); }

__device__ __forceinline__ void jit_col_seed(const double *col, int lane, double x60, v4d (&z)[4])
{
#ifdef JIT_ABL_NOSEED      // timing experiment (results are garbage): no column reads, no dependent multiplies in front of the first MFMAs
#pragma unroll
   for (int i = 0; i < 4; i++) z[i] = (v4d){x60, x60, x60, x60};
   return;
#endif
   const double2 *pc = (const double2 *)(col + (lane >> 4) * 16);
#pragma unroll
   for (int i = 0; i < 8; i++) {
      const double2 c = pc[i];
      z[i >> 1][(2 * i) & 3] = c.x * x60;
      z[i >> 1][(2 * i + 1) & 3] = c.y * x60;
   }
}

// ---- 61 states without the row padding (JIT_ROWTAIL; pmat_kernel layout 3) ---------------------------------------------------------
// Rows 0..47 of P are three 16 x 16 x 4 row blocks; rows 48..59 go through v_mfma_f64_4x4x4 (three row quartets m' = 12, 13, 14:
// an instruction does the quartet's 4 x 4 block of one k-block for all sixteen patterns in a quarter of a 16 x 16 x 4's pipe
// time, its result lands in the lanes' element m' — the partial's own layout, as in the 20-state kernel), row 60 is a dot product
// on the vector pipe, rows 61..63 do not exist: 45 + 45 / 4 = 56.25 instead of 60 big-instruction times per product.
// In the operand block the fourth row block's 1 KB slot of every k-block pair holds instead, for e = 0, 1 and m' = 12, 13, 14,
// the sixteen words [k][i] = P[4 m' + i][4 (2 kb2 + e) + k] at doubles (e * 3 + m' - 12) * 16 .., and from double 96 on
// [q][e] = P[60][4 (2 kb2 + e) + q] (column 60 itself arrives through the rank-1 seed).
#ifdef JIT_ROWTAIL
#define JIT_RT 1
#else
#define JIT_RT 0
#endif
struct JitRowTail {
   double a4[3];      // the k-block's three 4 x 4 x 4 operands (m' = 12, 13, 14); the pair's second k-block re-uses the registers
   double r60;        // the lane's share of row 60
};
__device__ __forceinline__ unsigned jit_lds_addr(const double *p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)p; }
template <int OFF>
__device__ __forceinline__ double jit_lds64(unsigned addr)
{
   double v;
   asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
   return v;
}
// request the three small operands of k-block 2 KB2 + E (base = LDS byte address of the block + the lane's (k, i) word)
template <int KB2, int E>
__device__ __forceinline__ void jit_rt_fetch(unsigned base, JitRowTail &rt)
{
   constexpr int S = (KB2 * 4 + 3) * 1024 + E * 384;
   rt.a4[0] = jit_lds64<S + 0 * 128>(base); rt.a4[1] = jit_lds64<S + 1 * 128>(base); rt.a4[2] = jit_lds64<S + 2 * 128>(base);
}
__device__ __forceinline__ void jit_rt_wait(JitRowTail &rt)      // the operands have arrived (LDS returns in order); nothing that uses them moves above
{
   asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rt.a4[0]), "+v"(rt.a4[1]), "+v"(rt.a4[2]));
}
#define JIT_MFMA4(A, B, C) __builtin_amdgcn_mfma_f64_4x4x4f64((A), (B), (C), 0, 0, 0)

// RB row blocks of 16 and KB k-blocks of 4 cover the model's states (4, 16 for 61; 2, 5 for 20): blocks beyond them are
// zero padding in P and are neither fetched nor multiplied; accumulators of skipped row blocks stay 0.
template <bool TAIL61 = false, int RB = 4, int KB = 16, class SIDE = JitNoSide>
__device__ __forceinline__ void jit_matvec(const double *sPbuf, int lane, const v4d (&x)[4], v4d (&y)[4], SIDE side = SIDE(),
                                           const double *col = nullptr, double x60 = 0)
{
   const double2 *sp = (const double2 *)sPbuf;
   v4d z[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
   if constexpr (TAIL61) jit_col_seed(col, lane, x60, z);
   double2 af[2][4];
   constexpr int KB2 = (KB + 1) / 2;
   constexpr bool RT = JIT_RT && TAIL61 && RB == 4 && KB == 16;      // rows 48..60 without the padding (see JitRowTail)
   constexpr int RBM = RT ? 3 : RB;
   JitRowTail rt;
   const unsigned rtbase = jit_lds_addr(sPbuf) + (((lane >> 4) << 2) + (lane & 3)) * 8;
   const double2 *sp60 = (const double2 *)(sPbuf + 3 * 128 + 96) + (lane >> 4);      // (+ kb2 * 256 double2: the pair's slot)
#pragma unroll
   for (int jb = RB; jb < 4; jb++) y[jb] = (v4d){0, 0, 0, 0};
   if constexpr (RT) { y[3] = z[3]; rt.r60 = 0; }
#pragma unroll
   for (int jb = 0; jb < RBM; jb++) af[0][jb] = sp[jb * 64 + lane];
#define JIT_RT_STEP(KB2V)                                                                                              \
   if constexpr (RT) {                                                                                                \
      const double2 p60 = sp60[(KB2V) * 256];                                                                         \
      jit_rt_fetch<KB2V, 0>(rtbase, rt);                                                                              \
      rt.r60 = fma(p60.x, x[(2 * (KB2V)) >> 2][(2 * (KB2V)) & 3], rt.r60);                                            \
      if ((KB2V) != 7) rt.r60 = fma(p60.y, x[(2 * (KB2V) + 1) >> 2][(2 * (KB2V) + 1) & 3], rt.r60);                   \
   }
#define JIT_RT_FETCH1(KB2V) if constexpr (RT) { jit_rt_fetch<KB2V, 1>(rtbase, rt); }
#define JIT_RT_SW(M)                                                                                                   \
   switch (kb2) {                                                                                                     \
   case 0: M(0) break; case 1: M(1) break; case 2: M(2) break; case 3: M(3) break;                                    \
   case 4: M(4) break; case 5: M(5) break; case 6: M(6) break; default: M(7) break;                                   \
   }
#pragma unroll
   for (int kb2 = 0; kb2 < KB2; kb2++) {
      if (kb2 + 1 < KB2) {
#pragma unroll
         for (int jb = 0; jb < RBM; jb++) af[(kb2 + 1) & 1][jb] = sp[((kb2 + 1) * 4 + jb) * 64 + lane];
      }
      if constexpr (RT) {      // (a literal pair index for the asm offsets)
         switch (kb2) {
         case 0: JIT_RT_STEP(0) break; case 1: JIT_RT_STEP(1) break; case 2: JIT_RT_STEP(2) break; case 3: JIT_RT_STEP(3) break;
         case 4: JIT_RT_STEP(4) break; case 5: JIT_RT_STEP(5) break; case 6: JIT_RT_STEP(6) break; default: JIT_RT_STEP(7) break;
         }
      }
      __builtin_amdgcn_sched_barrier(0);
      const double b0 = x[(2 * kb2) >> 2][(2 * kb2) & 3], b1 = x[(2 * kb2 + 1) >> 2][(2 * kb2 + 1) & 3];
      if (kb2 == 0) {
#pragma unroll
         for (int jb = 0; jb < RBM; jb++)
            y[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][jb].x, b0, z[jb], 0, 0, 0);
      }
      else {
#pragma unroll
         for (int jb = 0; jb < RBM; jb++) y[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kb2 & 1][jb].x, b0, y[jb], 0, 0, 0);
      }
      if constexpr (RT) {
         jit_rt_wait(rt);
         y[3].x = JIT_MFMA4(rt.a4[0], b0, y[3].x); y[3].y = JIT_MFMA4(rt.a4[1], b0, y[3].y); y[3].z = JIT_MFMA4(rt.a4[2], b0, y[3].z);
         if (kb2 != 7) { JIT_RT_SW(JIT_RT_FETCH1) }      // (sources are read at issue: the registers take the pair's second k-block)
      }
      side(kb2);
      if (!(TAIL61 && kb2 == 7) && 2 * kb2 + 1 < KB) {
#pragma unroll
         for (int jb = 0; jb < RBM; jb++) y[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kb2 & 1][jb].y, b1, y[jb], 0, 0, 0);
         if constexpr (RT) {
            jit_rt_wait(rt);
            y[3].x = JIT_MFMA4(rt.a4[0], b1, y[3].x); y[3].y = JIT_MFMA4(rt.a4[1], b1, y[3].y); y[3].z = JIT_MFMA4(rt.a4[2], b1, y[3].z);
         }
      }
      __builtin_amdgcn_sched_barrier(0);
   }
   if constexpr (RT) {      // row 60: the four state-quarter lanes' shares, onto the q = 0 lane's element 15 (states 61..63 stay 0)
      double r = rt.r60;
      r += __shfl_xor(r, 16);
      r += __shfl_xor(r, 32);
      y[3].w = lane < 16 ? y[3].w + r : 0.0;
   }
}

// jit_matvec with the next cherry's tip step folded in: y = P x as above, and t = tipA[ca] * tipB[cb] (the SET_TIP2 that
// follows in the program) gathered from LDS under the second half of the MFMAs, where the matrix pipe hides the
// ds_read_b128 traffic and its bank conflicts.  The two tip tables are the ring blocks after P; they only have to be
// resident by the midpoint MID = KB2 / 2, so MIDWAIT (outstanding vector-memory ops allowed there) + a barrier sit at
// kb2 == MID (refill pieces handed in through `side` are issued in the iterations before it).  The NP pieces of each tip
// row are fetched PPI per iteration from MID on, their products formed one iteration later.
template <int MIDWAIT, bool TAIL61 = false, int RB = 4, int KB = 16, class SIDE = JitNoSide>
__device__ __forceinline__ void jit_matvec_tip2(const double *sPbuf, int lane, const v4d (&x)[4], v4d (&y)[4], const double *ta, int ca,
                                                const double *tb, int cb, int q, v4d (&t)[4], SIDE side = SIDE(), const double *col = nullptr,
                                                double x60 = 0, JitAmb amb = JitAmb())
{
   (void)amb;
#ifdef JIT_AMB_OVERFLOW
   const int ca_full = ca, cb_full = cb;      // (the rows gathered under the MFMAs are those of the codes & 63; lanes with a code beyond them redo theirs below)
   ca &= 63; cb &= 63;
#endif
   constexpr int KB2 = (KB + 1) / 2, NP = KB2, MID = KB2 / 2, GI = KB2 - MID, PPI = (NP + GI - 1) / GI;
   const double2 *sp = (const double2 *)sPbuf;
   v4d z[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
   if constexpr (TAIL61) jit_col_seed(col, lane, x60, z);
   double2 af[2][4];
   constexpr bool RT = JIT_RT && TAIL61 && RB == 4 && KB == 16;      // rows 48..60 without the padding (see JitRowTail)
   constexpr int RBM = RT ? 3 : RB;
   JitRowTail rt;
   const unsigned rtbase = jit_lds_addr(sPbuf) + (((lane >> 4) << 2) + (lane & 3)) * 8;
   const double2 *sp60 = (const double2 *)(sPbuf + 3 * 128 + 96) + (lane >> 4);
   const int rowa = ca * 4 + q, rowb = cb * 4 + q, swa = TIP_SWZ(rowa), swb = TIP_SWZ(rowb);
   const char *pa = (const char *)ta + rowa * 128, *pb = (const char *)tb + rowb * 128;
   // (row-tail form: the tip rows fetched in an iteration are multiplied at its end — two more small-MFMA groups per iteration hide
   //  the LDS latency that the four-row-block form hides by deferring the products to the next iteration — half the staging registers)
   constexpr int TB = (JIT_RT && TAIL61 && RB == 4 && KB == 16) ? 1 : 2;
   double2 tv[TB][PPI], tw[TB][PPI];
#pragma unroll
   for (int jb = RB; jb < 4; jb++) y[jb] = (v4d){0, 0, 0, 0};
#pragma unroll
   for (int jb = 0; jb < 4; jb++) t[jb] = (v4d){0, 0, 0, 0};
   if constexpr (RT) { y[3] = z[3]; rt.r60 = 0; }
#pragma unroll
   for (int jb = 0; jb < RBM; jb++) af[0][jb] = sp[jb * 64 + lane];
#pragma unroll
   for (int kb2 = 0; kb2 < KB2; kb2++) {
      if (kb2 == MID) {
         asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MIDWAIT) : "memory");
         JIT_SYNC();
      }
      if (kb2 + 1 < KB2) {
#pragma unroll
         for (int jb = 0; jb < RBM; jb++) af[(kb2 + 1) & 1][jb] = sp[((kb2 + 1) * 4 + jb) * 64 + lane];
      }
      if constexpr (RT) {
         switch (kb2) {
         case 0: JIT_RT_STEP(0) break; case 1: JIT_RT_STEP(1) break; case 2: JIT_RT_STEP(2) break; case 3: JIT_RT_STEP(3) break;
         case 4: JIT_RT_STEP(4) break; case 5: JIT_RT_STEP(5) break; case 6: JIT_RT_STEP(6) break; default: JIT_RT_STEP(7) break;
         }
      }
      if (kb2 >= MID) {
#pragma unroll
         for (int e = 0; e < PPI; e++) {
            const int p = PPI * (kb2 - MID) + e;
            if (p < NP) {
               tv[kb2 & (TB - 1)][e] = *(const double2 *)(pa + ((p ^ swa) * 16));
               tw[kb2 & (TB - 1)][e] = *(const double2 *)(pb + ((p ^ swb) * 16));
            }
         }
      }
      __builtin_amdgcn_sched_barrier(0);
      const double b0 = x[(2 * kb2) >> 2][(2 * kb2) & 3], b1 = x[(2 * kb2 + 1) >> 2][(2 * kb2 + 1) & 3];
      if (kb2 == 0) {
#pragma unroll
         for (int jb = 0; jb < RBM; jb++)
            y[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][jb].x, b0, z[jb], 0, 0, 0);
      }
      else {
#pragma unroll
         for (int jb = 0; jb < RBM; jb++) y[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kb2 & 1][jb].x, b0, y[jb], 0, 0, 0);
      }
      if constexpr (RT) {
         jit_rt_wait(rt);
         y[3].x = JIT_MFMA4(rt.a4[0], b0, y[3].x); y[3].y = JIT_MFMA4(rt.a4[1], b0, y[3].y); y[3].z = JIT_MFMA4(rt.a4[2], b0, y[3].z);
         if (kb2 != 7) { JIT_RT_SW(JIT_RT_FETCH1) }      // (sources are read at issue: the registers take the pair's second k-block)
      }
      if (TB == 2 && kb2 > MID) {      // products of the rows fetched one iteration ago
#pragma unroll
         for (int e = 0; e < PPI; e++) {
            const int p = PPI * (kb2 - 1 - MID) + e;
            if (p < NP) {
               t[p >> 1][(2 * p) & 3] = tv[(kb2 - 1) & 1][e].x * tw[(kb2 - 1) & 1][e].x;
               t[p >> 1][(2 * p + 1) & 3] = tv[(kb2 - 1) & 1][e].y * tw[(kb2 - 1) & 1][e].y;
            }
         }
      }
      side(kb2);
      if (!(TAIL61 && kb2 == 7) && 2 * kb2 + 1 < KB) {
#pragma unroll
         for (int jb = 0; jb < RBM; jb++) y[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kb2 & 1][jb].y, b1, y[jb], 0, 0, 0);
         if constexpr (RT) {
            jit_rt_wait(rt);
            y[3].x = JIT_MFMA4(rt.a4[0], b1, y[3].x); y[3].y = JIT_MFMA4(rt.a4[1], b1, y[3].y); y[3].z = JIT_MFMA4(rt.a4[2], b1, y[3].z);
         }
      }
      if (TB == 1 && kb2 >= MID) {      // products of the rows fetched at the top of this iteration
#pragma unroll
         for (int e = 0; e < PPI; e++) {
            const int p = PPI * (kb2 - MID) + e;
            if (p < NP) {
               t[p >> 1][(2 * p) & 3] = tv[0][e].x * tw[0][e].x;
               t[p >> 1][(2 * p + 1) & 3] = tv[0][e].y * tw[0][e].y;
            }
         }
      }
      __builtin_amdgcn_sched_barrier(0);
   }
   if constexpr (RT) {
      double r = rt.r60;
      r += __shfl_xor(r, 16);
      r += __shfl_xor(r, 32);
      y[3].w = lane < 16 ? y[3].w + r : 0.0;
   }
#pragma unroll
   for (int e = 0; e < PPI && TB == 2; e++) {
      const int p = PPI * (KB2 - 1 - MID) + e;
      if (p < NP) {
         t[p >> 1][(2 * p) & 3] = tv[(KB2 - 1) & 1][e].x * tw[(KB2 - 1) & 1][e].x;
         t[p >> 1][(2 * p + 1) & 3] = tv[(KB2 - 1) & 1][e].y * tw[(KB2 - 1) & 1][e].y;
      }
   }
#ifdef JIT_AMB_OVERFLOW
   if (ca_full >= 64 || cb_full >= 64) {      // (both tables are still resident: the ring's refills go to slots of blocks consumed before this step)
      double2 va[8], vb[8];
      if (ca_full >= 64) jit_tip_overflow<NP>(amb, ta, ca_full, q, va);
      else {
#pragma unroll
         for (int p = 0; p < 8; p++) va[p] = p < NP ? *(const double2 *)(pa + ((p ^ swa) * 16)) : make_double2(0.0, 0.0);
      }
      if (cb_full >= 64) jit_tip_overflow<NP>(amb, tb, cb_full, q, vb);
      else {
#pragma unroll
         for (int p = 0; p < 8; p++) vb[p] = p < NP ? *(const double2 *)(pb + ((p ^ swb) * 16)) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int p = 0; p < 8; p++) { t[p >> 1][(2 * p) & 3] = va[p].x * vb[p].x; t[p >> 1][(2 * p + 1) & 3] = va[p].y * vb[p].y; }
   }
#endif
}

// Partials of stack slots beyond the register arrays live in global scratch (trees with a deep partial stack): one coalesced
// 512-byte store / load per register of the wave.  These are ordinary (compiler-visible) memory operations; the counted waits of
// the operand ring stay valid because loads complete in order and a wait that also covers younger operations only waits longer.
#define JIT_SPILL_PTR(K) (a.stack_scratch + (((long)blockIdx.x * a.stack_overflow_slots + (K)) * JIT_WAVES + wave) * 1024 + lane)
__device__ __forceinline__ void jit_spill(const v4d (&y)[4], double *sp)
{
#pragma unroll
   for (int jb = 0; jb < 4; jb++) { sp[(jb * 4 + 0) * 64] = y[jb].x; sp[(jb * 4 + 1) * 64] = y[jb].y; sp[(jb * 4 + 2) * 64] = y[jb].z; sp[(jb * 4 + 3) * 64] = y[jb].w; }
}
__device__ __forceinline__ void jit_mul_mem(v4d (&y)[4], const double *sp)   // y = (spilled s) * y
{
#pragma unroll
   for (int jb = 0; jb < 4; jb++) {
      y[jb].x *= sp[(jb * 4 + 0) * 64]; y[jb].y *= sp[(jb * 4 + 1) * 64]; y[jb].z *= sp[(jb * 4 + 2) * 64]; y[jb].w *= sp[(jb * 4 + 3) * 64];
   }
}

// (JIT_STORE_MODE, experiments through PAML_AMD_JIT_STORE: 0 = nontemporal, 1 = plain, 2 = no store at all — timing only, the resident partials stay unwritten)
#ifndef JIT_STORE_MODE
#define JIT_STORE_MODE 0
#endif
#if JIT_STORE_MODE == 0
#define JIT_STORE16(V, P) __builtin_nontemporal_store((V), (P))
#elif JIT_STORE_MODE == 1
#define JIT_STORE16(V, P) (*(P) = (V))
#else
#define JIT_STORE16(V, P) ((void)0)
#endif
// STORE / LOAD of a resident partial (keep-partials mode; layout: part_load / part_store above — element m of the partial is y[m >> 2][m & 3],
// a lane's elements 2 i, 2 i + 1 one 16-byte access, a wave instruction 1 KB).  Compiler-visible memory operations, like the spills.
// (streamed past the caches: 7 GB per evaluation at the benchmark's size would otherwise push the P(t) blocks and tip tables, re-read by
//  every tile, out of L2)
__device__ __forceinline__ void jit_store(const v4d (&y)[4], double *p, int lane)
{
   part2_t *p2 = (part2_t *)p + lane;
#pragma unroll
   for (int i = 0; i < 8; i++) JIT_STORE16(((part2_t){y[i >> 1][(2 * i) & 3], y[i >> 1][(2 * i + 1) & 3]}), p2 + i * 64);
}
__device__ __forceinline__ void jit_load(v4d (&y)[4], const double *p, int lane)
{
   const part2_t *p2 = (const part2_t *)p + lane;
#pragma unroll
   for (int i = 0; i < 8; i++) { const part2_t v = p2[i * 64]; y[i >> 1][(2 * i) & 3] = v.x; y[i >> 1][(2 * i + 1) & 3] = v.y; }
}
#define JIT_STORE_PIECE(Y, PTR, I) JIT_STORE16(((part2_t){Y[(I) >> 1][(2 * (I)) & 3], Y[(I) >> 1][(2 * (I) + 1) & 3]}), (part2_t *)(PTR) + lane + (I)*64)
#define JIT_PART_PTR(NODE) (a.partials + (((long)iclass * a.n_int + ((NODE) - a.n_tips)) * a.part_groups + tg0 + wave) * 1024)
#define JIT_PART_DST(NODE) (wave_in ? JIT_PART_PTR(NODE) : a.part_dump + wave * 1024)

__device__ __forceinline__ void jit_mul(v4d (&y)[4], const v4d (&s)[4])   // y = s * y  (codeml.c:3573)
{
#pragma unroll
   for (int jb = 0; jb < 4; jb++) y[jb] = s[jb] * y[jb];
}

__device__ __forceinline__ void jit_init_ones(v4d (&y)[4], int q, int n)
{
#pragma unroll
   for (int m = 0; m < 16; m++) y[m >> 2][m & 3] = (4 * m + q < n) ? 1.0 : 0.0;
}

__device__ __forceinline__ void jit_init_tip(v4d (&y)[4], int code, int q, int cleandata)
{
#pragma unroll
   for (int m = 0; m < 16; m++) y[m >> 2][m & 3] = (cleandata && 4 * m + q == code) ? 1.0 : 0.0;
}

// Tip factors: NP = number of 16-byte pieces (two states each) of a table row that carry states of the model (8 at 61
// states, 3 at 20); the rest of the row is zero padding and is not read.
template <int NP>
__device__ __forceinline__ void tip_lds_n(const double *tab, int code, int q, double2 (&v)[8], JitAmb amb = JitAmb())
{
   (void)amb;
   const int row = JIT_FASTCODE(code) * 4 + q, swz = TIP_SWZ(row);
   const char *base = (const char *)tab + row * 128;
#pragma unroll
   for (int p = 0; p < 8; p++) v[p] = p < NP ? *(const double2 *)(base + ((p ^ swz) * 16)) : make_double2(0.0, 0.0);
#ifdef JIT_AMB_OVERFLOW
   if (code >= 64) jit_tip_overflow<NP>(amb, tab, code, q, v);
#endif
}

template <int NP = 8>
__device__ __forceinline__ void jit_tip_set(v4d (&y)[4], const double *tab, int code, int q, int lane, JitAmb amb = JitAmb())
{
   double2 v[8];
   tip_lds_n<NP>(tab, code, q, v, amb);
#pragma unroll
   for (int i = 0; i < 8; i++) { y[i >> 1][(2 * i) & 3] = v[i].x; y[i >> 1][(2 * i + 1) & 3] = v[i].y; }
}

template <int NP = 8>
__device__ __forceinline__ void jit_tip_mul(v4d (&y)[4], const double *tab, int code, int q, int lane, JitAmb amb = JitAmb())
{
   double2 v[8];
   tip_lds_n<NP>(tab, code, q, v, amb);
#pragma unroll
   for (int i = 0; i < NP; i++) { y[i >> 1][(2 * i) & 3] *= v[i].x; y[i >> 1][(2 * i + 1) & 3] *= v[i].y; }
}

template <int NP = 8>
__device__ __forceinline__ void jit_tip2_set(v4d (&y)[4], const double *ta, int ca, const double *tb, int cb, int q, int lane, JitAmb amb = JitAmb())
{
   double2 v[8], w[8];
   tip_lds_n<NP>(ta, ca, q, v, amb);
   tip_lds_n<NP>(tb, cb, q, w, amb);
#pragma unroll
   for (int i = 0; i < 8; i++) { y[i >> 1][(2 * i) & 3] = v[i].x * w[i].x; y[i >> 1][(2 * i + 1) & 3] = v[i].y * w[i].y; }
}

template <int NP = 8>
__device__ __forceinline__ void jit_tip2_mul(v4d (&y)[4], const double *ta, int ca, const double *tb, int cb, int q, int lane, JitAmb amb = JitAmb())
{
   double2 v[8], w[8];
   tip_lds_n<NP>(ta, ca, q, v, amb);
   tip_lds_n<NP>(tb, cb, q, w, amb);
#pragma unroll
   for (int i = 0; i < NP; i++) {
      y[i >> 1][(2 * i) & 3] = (y[i >> 1][(2 * i) & 3] * v[i].x) * w[i].x;
      y[i >> 1][(2 * i + 1) & 3] = (y[i >> 1][(2 * i + 1) & 3] * v[i].y) * w[i].y;
   }
}

__device__ __forceinline__ double jit_scale(v4d (&y)[4], int q, int n)   // NodeScale, treesub.c:7200-7230
{
   double mx = 0;
#pragma unroll
   for (int m = 0; m < 16; m++) mx = y[m >> 2][m & 3] > mx ? y[m >> 2][m & 3] : mx;
   double o = __shfl_xor(mx, 16);
   mx = o > mx ? o : mx;
   o = __shfl_xor(mx, 32);
   mx = o > mx ? o : mx;
   if (mx < 1e-300) {
      jit_init_ones(y, q, n);
      return -800;
   }
#pragma unroll
   for (int m = 0; m < 16; m++) y[m >> 2][m & 3] /= mx;
   return log(mx);
}

// Root stage with pi and the weight flag already in LDS (no vector-memory loads whose wait would drain the ring's DMAs).
__device__ __forceinline__ void jit_root_lds(const PruneArgs &a, const v4d (&x)[4], double lnscale, const double *spi, int flag, int iclass,
                                             int q, int h, bool valid)
{
   const double *pq = spi + q * 16;
   double f = 0;
#pragma unroll
   for (int m = 0; m < 16; m++) f = fma(pq[m], x[m >> 2][m & 3], f);
   f += __shfl_xor(f, 16);
   f += __shfl_xor(f, 32);
   if (q == 0 && valid) {
      // fx_r treesub.c:7731-7749 / lfun 7782-7798: the floor here, log + scale factors in the reduction kernel
      if (f <= 0) f = (a.mode == PAML_AMD_MODE_LFUN ? 1e-80 : 1e-300);
#ifdef JIT_NT_STORE      // experiment: the class likelihoods streamed past L2 (nothing dirty left for the end-of-kernel release)
      __builtin_nontemporal_store(flag ? f : 0.0, a.fhK + (long)iclass * a.n_patt + h);
#else
      a.fhK[(long)iclass * a.n_patt + h] = flag ? f : 0.0;
#endif
      if (a.n_scale) a.fscale[(long)iclass * a.n_patt + h] = lnscale;
   }
}

// Explicit counted wait for the LDS-DMA stream (loads retire in issue order): at most N vector-memory operations of this
// wave may still be in flight afterwards.
#define JIT_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// a never-taken uniform branch: ends a basic block of the per-tree kernel's straight-line code (jit.h, split_mode)
#define JIT_SPLIT() if (__builtin_expect(a.n_tiles == 0x7fffffff, 0)) asm volatile("s_trap 2");

// ---- building blocks of the specialised one-pattern-per-lane kernels (4 / 5 / 20 states; jit.h: jit_generate_valu) ----
// Same arithmetic as prune_valu<N>: P(t) entries are wave-uniform and come through the constant address space (s_load ->
// SGPR operands of v_fma_f64); with the walk unrolled the compiler issues those loads far ahead of their use and the
// partial stack lives in renamed registers instead of an indexed (scratch) array.
template <int N>
__device__ __forceinline__ void jv_matvec(const double *Pg, const double (&x)[N], double (&y)[N])
{
   const CONST_AS double *P = as_const(Pg);
#pragma unroll
   for (int j = 0; j < N; j++) {
      double t = 0;
#pragma unroll
      for (int k = 0; k < N; k++) t = fma(P[j * N + k], x[k], t);
      y[j] = t;
   }
}

template <int N>
__device__ __forceinline__ double jv_scale(double (&x)[N])      // NodeScale treesub.c:7200-7230
{
   double mx = 0;
#pragma unroll
   for (int j = 0; j < N; j++) mx = x[j] > mx ? x[j] : mx;
   if (mx < 1e-300) {
#pragma unroll
      for (int j = 0; j < N; j++) x[j] = 1.0;
      return -800;
   }
#pragma unroll
   for (int j = 0; j < N; j++) x[j] /= mx;
   return log(mx);
}

template <int N>
__device__ __forceinline__ void jv_root(const PruneArgs &a, const double (&x)[N], double lnscale, int gene, int iclass, int h, bool valid)
{
   const CONST_AS double *pi = as_const(a.pi + (long)(a.n_pi > 1 ? gene : 0) * N);
   double f = 0;
#pragma unroll
   for (int j = 0; j < N; j++) f = fma(pi[j], x[j], f);
   if (valid) {
      double out = 0;
      if (a.weights[h] > 0) out = root_value(a, f, lnscale);
      a.fhK[(long)iclass * a.n_patt + h] = out;
   }
}

// ---- the reduction's tail, shared by every kernel that forms partial sums ---------------------------------------------
// Fixed-order total of nb partial sums by one 256-thread workgroup (what reduce_stage2 does): lane sums over i, i + 256, ...
// then the butterfly and the four waves.  `coherent`: the partials were written by other workgroups of the SAME launch —
// read them past the (non-coherent) L1.
template <int GROUPS = 1>
__device__ __forceinline__ double red_total256(const double *partial, int nb, bool coherent, double *sw4)
{
   double acc = 0;
   if (GROUPS == 1 || threadIdx.x < 256)      // (GROUPS > 1: a workgroup of 256 x GROUPS threads; the first 256 do the sums,
      for (int i = threadIdx.x; i < nb; i += 256) {      //  all of them take part in the barriers)
         double v;
         if (coherent) {
            const unsigned long long u = __hip_atomic_load((const unsigned long long *)(partial + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __longlong_as_double((long long)u);
         }
         else v = partial[i];
         acc += v;
      }
#pragma unroll
   for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
   __syncthreads();
   if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) sw4[threadIdx.x >> 6] = acc;
   __syncthreads();
   return (sw4[0] + sw4[1]) + (sw4[2] + sw4[3]);
}

// A workgroup's weighted sum -> its slot of the global partial array; with a counter, the workgroup that finishes last also
// forms the total (single GPU: no separate stage-2 launch).  The order of every sum is fixed, so the result is deterministic.
// "Last" is found with two levels of tickets — groups of RED_TICKET_GROUP workgroups, then the groups — so that a thousand
// workgroups finishing together do not queue on one L2 atomic (measured: 35 us for 977 tickets on a single counter).
// counter[0] = groups done, counter[1 + g] = workgroups of group g done; all return to zero for the next launch.
// (GROUPS > 1: a workgroup of 256 x GROUPS threads whose first 256 hold the sums; the others only take part in the barriers.)
#define RED_TICKET_GROUP 32
#define RED_TICKET_WORDS 40      /* ints per batch element: 1 + ceil(1024 / 32) groups, padded */
template <int GROUPS = 1>
__device__ __forceinline__ void red_block_finish(double acc, double *partial_row, int slot, int nb_total, double *out, int *counter)
{
   __shared__ double sw[4];
   __shared__ int s_last;
#pragma unroll
   for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
   if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) sw[threadIdx.x >> 6] = acc;
   __syncthreads();
   if (threadIdx.x == 0) {
      const double part = (sw[0] + sw[1]) + (sw[2] + sw[3]);
      int last = 0;
      if (counter) {
         // The partial sum has to be visible to whichever workgroup ends up last, possibly on another XCD (each XCD has its own
         // L2).  A __threadfence() would do it — by writing back EVERY dirty line of this XCD's L2 (the 8 MB of class likelihoods
         // the kernel has just stored): 35 us for a 5 us kernel.  Instead the one value goes out as an agent-scope store (written
         // through), its completion is awaited, and only then is the ticket drawn; the reader uses agent-scope loads.
         const int nb = (int)gridDim.x, g = (int)blockIdx.x / RED_TICKET_GROUP, ng = (nb + RED_TICKET_GROUP - 1) / RED_TICKET_GROUP;
         const int gsize = min(RED_TICKET_GROUP, nb - g * RED_TICKET_GROUP);
         __hip_atomic_store((unsigned long long *)(partial_row + slot), (unsigned long long)__double_as_longlong(part), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         if (ng > RED_TICKET_WORDS - 1)      // (more groups than ticket words: one level)
            last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1;
         else if (__hip_atomic_fetch_add(counter + 1 + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
            __hip_atomic_store(counter + 1 + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1;
         }
      }
      else {
         partial_row[slot] = part;
         if (nb_total == 1) *out = part;      // a single block: its sum is the total (what the fixed-order pass over one value returns), no second launch
      }
      s_last = last;
   }
   __syncthreads();
   if (s_last) {
      const double tot = red_total256<GROUPS>(partial_row, nb_total, true, sw);
      if (threadIdx.x == 0) {
         *out = tot;
         __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
      }
   }
}

// ---- the cooperative per-tree kernel for small data sets (jit.h: jit_generate_coop) ---------------------------------------------
// prune_mfma64_coop (kernels_prune.h) unrolled for one tree: four waves share a 16-pattern group, wave w owns row block w of every
// product.  What the interpreter form waits for at every step — the op fetch, the hand-over of the staged P(t) block, the tip rows
// requested one step ahead — is resolved when the kernel is generated: every operand (a wave's quarter of a branch's P(t) in A-operand
// order: eight 16-byte words per lane; a wave's quarter of a tip's row: two) is requested straight into registers as far ahead as the
// register budget allows — for trees of ~15 taxa all of them at the kernel's start, one round trip to L2 for the whole walk — and a
// product step is: publish the quarter of x, one barrier, eight LDS reads, sixteen dependent MFMAs.  Same accumulation order per row
// block, same root sum, same reduction order as the interpreter kernels + reduce_stage1 / reduce_stage2: the same bits.
// The reduction is inside: the workgroup that finishes last for its batch element (tickets, as red_block_finish) forms mixture + log +
// the fixed-order chunk sums and the total — ONE launch after P(t) per evaluation.
__device__ __forceinline__ double coopj_ld(const double *p)      // a value another workgroup of this launch wrote (agent scope: past the L2 of this XCD)
{
   return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void coopj_st(double *p, double v)
{
   __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// pattern_lnf of kernels_reduce.h on class likelihoods other workgroups of this launch have just written.  The coherent loads go out
// eight at a time (one after the other they are a chain of K round trips past the L2: M8's eleven classes), the sums keep the class order.
__device__ __forceinline__ double coopj_pattern_lnf(const PruneArgs &a, const double *fhK, const double *freqK, int h)
{
   if (a.mode == PAML_AMD_MODE_LFUN) return coopj_ld(fhK + h);
   const int K = a.Km;
   double fh = 0;
   if (a.n_scale) {      // log-sum-exp around the first maximum (treesub.c:7640-7649)
      double t = coopj_ld(fhK + h);
      for (int i0 = 1; i0 < K; i0 += 8) {
         double v[8];
#pragma unroll
         for (int j = 0; j < 8; j++) v[j] = i0 + j < K ? coopj_ld(fhK + (long)(i0 + j) * a.n_patt + h) : t;
#pragma unroll
         for (int j = 0; j < 8; j++) if (i0 + j < K && v[j] > t) t = v[j];
      }
      for (int i0 = 0; i0 < K; i0 += 8) {
         double v[8];
#pragma unroll
         for (int j = 0; j < 8; j++) v[j] = i0 + j < K ? coopj_ld(fhK + (long)(i0 + j) * a.n_patt + h) : t;
#pragma unroll
         for (int j = 0; j < 8; j++) if (i0 + j < K) fh += freqK[i0 + j] * exp(v[j] - t);
      }
      return t + log(fh);
   }
   for (int i0 = 0; i0 < K; i0 += 8) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = i0 + j < K ? coopj_ld(fhK + (long)(i0 + j) * a.n_patt + h) : 0.0;
#pragma unroll
      for (int j = 0; j < 8; j++) if (i0 + j < K) fh += freqK[i0 + j] * v[j];
   }
   if (fh <= 0) fh = 1e-300;
   return log(fh);
}
// wg: this workgroup's number among the n_wg of its batch element `bat`.  Every workgroup of the launch comes here (also those whose
// 16 patterns lie past the end of their gene), after its class likelihoods have been stored with coopj_st.
__device__ __forceinline__ void coopj_finish(const PruneArgs &a, int bat, int wg, int n_wg)
{
   __shared__ double cj_sw[4];
   __shared__ int cj_last;
   __shared__ double cj_part[1024];
   const int tid = threadIdx.x;
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this lane's class likelihoods are out (written through) ...
   __syncthreads();                                       // ... and so are the workgroup's
   int *counter = a.red_counter + bat * RED_TICKET_WORDS;
   if (tid == 0) {      // two levels of tickets (red_block_finish): groups of RED_TICKET_GROUP workgroups, then the groups
      const int g = wg / RED_TICKET_GROUP, ng = (n_wg + RED_TICKET_GROUP - 1) / RED_TICKET_GROUP, gsize = min(RED_TICKET_GROUP, n_wg - g * RED_TICKET_GROUP);
      int last = 0;
      if (ng > RED_TICKET_WORDS - 1) last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_wg - 1;
      else if (__hip_atomic_fetch_add(counter + 1 + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
         __hip_atomic_store(counter + 1 + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1;
      }
      cj_last = last;
   }
   __syncthreads();
   if (!cj_last) return;
   if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
   // reduce_stage1's loop over the chunks of this element (kernels_reduce.h), then reduce_stage2's total
   const double *fhK = a.fhK + (long)bat * a.Km * a.n_patt, *freqK = a.freqK + bat * a.freqK_bs;
   const int nb = a.nb_local;
   for (int c = 0; c < nb; c++) {
      const int lo = c * a.chunk, hi = min(a.n_patt, lo + a.chunk);
      double acc = 0;
      for (int h = lo + tid; h < hi; h += 256) {
         double v = 0;
         if (a.weights[h] > 0) {
            v = coopj_pattern_lnf(a, fhK, freqK, h);
            acc += v * a.weights[h];
         }
         if (a.lnf) a.lnf[(long)bat * a.n_patt + h] = v;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
      if ((tid & 63) == 0) cj_sw[tid >> 6] = acc;
      __syncthreads();
      if (tid == 0) {
         const double part = (cj_sw[0] + cj_sw[1]) + (cj_sw[2] + cj_sw[3]);
         if (c < 1024) cj_part[c] = part;
         a.red_partial[(long)bat * a.nb_stride + a.first_chunk + c] = part;
      }
      __syncthreads();
   }
   if (a.nb_stride == 1) {      // a single chunk: its sum is the total
      if (tid == 0) a.red_out[bat] = cj_part[0];
      return;
   }
   double acc = 0;
   for (int i = tid; i < a.nb_stride; i += 256) acc += (i >= a.first_chunk && i < a.first_chunk + nb && i - a.first_chunk < 1024) ? cj_part[i - a.first_chunk] : 0.0;
#pragma unroll
   for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
   __syncthreads();
   if ((tid & 63) == 0) cj_sw[tid >> 6] = acc;
   __syncthreads();
   if (tid == 0) a.red_out[bat] = (cj_sw[0] + cj_sw[1]) + (cj_sw[2] + cj_sw[3]);
}

#define COOPJ_PROLOGUE                                                                                             \
   __shared__ __attribute__((aligned(16))) double sX[2][1024];      /* the operand partial, [m >> 1][lane][m & 1] */ \
   __shared__ double sR[4][16];                                                                                 \
   const int tid = threadIdx.x, lane = tid & 63;                                                                \
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                   \
   const int q = lane >> 4, hl = lane & 15;                                                                     \
   const int sub = blockIdx.x & 3, tile = (blockIdx.x >> 2) % a.n_tiles, iclass = (blockIdx.x >> 2) / a.n_tiles; \
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y + 16 * sub;                       \
   const int hend = as_const(a.gene_off)[gene + 1];                                                             \
   const bool empty = h0 >= hend;                                                                               \
   const int h = h0 + hl;                                                                                       \
   const bool valid = h < hend;                                                                                 \
   const int hc = valid ? h : hend - 1;                                                                         \
   const long pset = (long)gene * a.K + iclass;                                                                 \
   const double *Pint = a.pint + pset * a.n_nodes * 4096;                                                       \
   const long tipstride = a.tip_words;                                                                          \
   const double *Ptip = a.ptip + pset * a.n_nodes * tipstride;                                                  \
   const int n = a.n;                                                                                           \
   double lnscale = 0;                                                                                          \
   (void)lnscale; (void)n; (void)sR; (void)Ptip; (void)Pint;
// a wave's quarter of the A operands of the branch above NODE (its row block of every k-block pair): 16-byte words per lane, the first
// COOPJ_NP pairs — the k-blocks that hold states of the model (8 pairs at 61 states, 3 at 20: the others are zero padding)
#ifndef COOPJ_NP
#define COOPJ_NP 8
#define COOPJ_KB 16
#endif
#define COOPJ_P(V, NODE)                                                                                          \
   double2 V[COOPJ_NP];                                                                                         \
   { const double2 *sp_ = (const double2 *)(Pint + (long)(NODE)*4096) + wave * 64 + lane;                        \
     _Pragma("unroll") for (int p_ = 0; p_ < COOPJ_NP; p_++) V[p_] = sp_[p_ * 256]; }
// a wave's quarter (elements m = 4 wave + r) of the row (code, q) of a tip's column table: pieces 2 wave, 2 wave + 1
#define COOPJ_T(V, TIP, CODE)                                                                                     \
   double2 V##a, V##b;                                                                                          \
   { const int row_ = (CODE)*4 + q, swz_ = TIP_SWZ(row_);                                                        \
     const double2 *pt_ = (const double2 *)(Ptip + (long)(TIP)*tipstride + row_ * 16);                          \
     V##a = pt_[(2 * wave) ^ swz_]; V##b = pt_[(2 * wave + 1) ^ swz_]; }
// y = (row block `wave` of P) . x: the quarter is published, one barrier, eight LDS reads, sixteen dependent MFMAs (k-blocks ascending
// into one accumulator: the order of mfma_matvec and prune_mfma64_coop)
#define COOPJ_MATVEC(PV, X, Y, XB)                                                                                \
   { double2 *xs_ = (double2 *)sX[XB];                                                                          \
     xs_[(2 * wave) * 64 + lane] = make_double2(X[0], X[1]);                                                    \
     xs_[(2 * wave + 1) * 64 + lane] = make_double2(X[2], X[3]);                                                \
     __syncthreads();                                                                                           \
     const double2 *xr_ = (const double2 *)sX[XB];                                                              \
     double2 xv_[COOPJ_NP];                                                                                     \
     _Pragma("unroll") for (int p_ = 0; p_ < COOPJ_NP; p_++) xv_[p_] = xr_[p_ * 64 + lane];                     \
     __builtin_amdgcn_sched_barrier(0);      /* all the reads in flight before the first MFMA: one LDS latency per product */ \
     v4d acc_ = {0, 0, 0, 0};                                                                                   \
     _Pragma("unroll") for (int k_ = 0; k_ < COOPJ_NP; k_++) {      /* (k-blocks of zero padding add + 0.0: left out) */ \
        acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(PV[k_].x, xv_[k_].x, acc_, 0, 0, 0);                        \
        if (2 * k_ + 1 < COOPJ_KB) acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(PV[k_].y, xv_[k_].y, acc_, 0, 0, 0); \
     }                                                                                                          \
     Y[0] = acc_[0]; Y[1] = acc_[1]; Y[2] = acc_[2]; Y[3] = acc_[3]; }
// NodeScale treesub.c:7200-7230: the maximum over all the states of the pattern = over the four waves' quarters
#define COOPJ_SCALE(X)                                                                                            \
   { double mx_ = 0;                                                                                            \
     _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) mx_ = X[r_] > mx_ ? X[r_] : mx_;                          \
     double o_ = __shfl_xor(mx_, 16); mx_ = o_ > mx_ ? o_ : mx_;                                                \
     o_ = __shfl_xor(mx_, 32); mx_ = o_ > mx_ ? o_ : mx_;                                                       \
     if (q == 0) sR[wave][hl] = mx_;                                                                            \
     __syncthreads();                                                                                           \
     _Pragma("unroll") for (int w2_ = 0; w2_ < 4; w2_++) { const double v_ = sR[w2_][hl]; mx_ = v_ > mx_ ? v_ : mx_; } \
     __syncthreads();                                                                                           \
     double fac_;                                                                                               \
     if (mx_ < 1e-300) { _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) X[r_] = (4 * (4 * wave + r_) + q < n) ? 1.0 : 0.0; fac_ = -800; } \
     else { _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) X[r_] /= mx_; fac_ = log(mx_); }                   \
     lnscale += fac_; }
// the root: the whole partial to wave 0, which sums it as the other kernels do (MFMA_ROOT_CASE) and stores fx_r's value
#define COOPJ_ROOT(X, XB)                                                                                         \
   { double2 *xs_ = (double2 *)sX[XB];                                                                          \
     xs_[(2 * wave) * 64 + lane] = make_double2(X[0], X[1]);                                                    \
     xs_[(2 * wave + 1) * 64 + lane] = make_double2(X[2], X[3]);                                                \
     __syncthreads();                                                                                           \
     if (wave == 0) {                                                                                           \
        const double2 *xr_ = (const double2 *)sX[XB];                                                           \
        const double *pq_ = a.pi + (long)(a.n_pi > 1 ? gene : 0) * 64 + q * 16;                                 \
        double f_ = 0;                                                                                          \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; p_++) { const double2 v_ = xr_[p_ * 64 + lane]; f_ = fma(pq_[2 * p_], v_.x, f_); f_ = fma(pq_[2 * p_ + 1], v_.y, f_); } \
        f_ += __shfl_xor(f_, 16);                                                                               \
        f_ += __shfl_xor(f_, 32);                                                                               \
        if (q == 0 && valid) {                                                                                  \
           double out_ = 0;                                                                                     \
           if (a.weights[h] > 0) out_ = root_value(a, f_, lnscale);                                             \
           coopj_st(a.fhK + (long)iclass * a.n_patt + h, out_);                                                 \
        }                                                                                                       \
     } }

// ---- fused one-pattern-per-lane kernel (4 / 5 states; jit.h: jit_generate_valu_fused) ---------------------------------------
// One workgroup owns one reduction chunk of patterns and walks it 256 patterns at a time.  Per pattern the tip codes are read
// ONCE (pattern-major, 4 codes per dword), the classes are the INNER loop — the class likelihoods never leave registers unless
// asked for — and the mixture, log, weight and the chunk's partial sum are formed in the same kernel.  Tip factors come from
// LDS tables (rows of the tips' P summed over each code's state set; for a cherry of two tips the products of their rows),
// filled once per workgroup from pmat's tables: gathering them from L2 instead made the texture path, not the FP64 pipe, the
// limit of the round-1 kernel (two 16-byte gathers per tip and lane against sixteen DFMAs).
template <int N>
__device__ __forceinline__ void jvf_row_set(double (&x)[N], const double *row)
{
#pragma unroll
   for (int j = 0; j < N; j++) x[j] = row[j];
}
template <int N>
__device__ __forceinline__ void jvf_row_mul(double (&x)[N], const double *row)
{
#pragma unroll
   for (int j = 0; j < N; j++) x[j] *= row[j];
}
#define JVF_CODE(T) ((int)((zw[(T) >> 2] >> (((T) & 3) * 8)) & 0xffu))

// ---- 20-state models on v_mfma_f64_4x4x4_4b_f64 (jit.h: jit_generate_m20) --------------------------------------------------------
// 20 = 5 blocks of 4 states: with the 4x4x4 instruction P . L has NO padding (25 products of 4 x 4 blocks per 16 patterns = 800
// FLOP per pattern, the algorithmic count), where v_mfma_f64_16x16x4 spends 10 instructions of which 37.5 % of the rows are
// zeros.  The instruction runs at the full FP64 rate with one wave per SIMD and its result layout is its B-operand layout
// (profiles/r02_mfma4_layout.txt), so a partial is FIVE doubles per lane — block m holds state 4 m + (lane >> 4) of pattern
// lane & 15 — and chains from node to node in registers.  The A operand of block (I, K) is P[4I + i][4K + k] in lane
// 16 k + 4 b + i: one ds_read_b64 from the row-major P(t) of the branch in LDS (sixteen distinct 8-byte words, each read by the
// four lanes of the same (i, k): conflict-free), shared by the wave's two pattern groups.
#define M20_MFMA(A, B, C) __builtin_amdgcn_mfma_f64_4x4x4f64((A), (B), (C), 0, 0, 0)
// The A blocks are fetched with ds_read_b64 issued from inline asm: left to the compiler, neighbouring reads are merged into
// ds_read2_b64, which moves 128 bytes per LDS cycle where ds_read_b64 moves 256 (MI355X_MICROARCH.md, LDS table) — with eight waves
// reading twenty-five blocks per product plus the tip rows, the merged form keeps the LDS busy for as long as the MFMAs take.
// Column K + 1 is requested before column K's MFMAs; LDS returns in order, so "at most five LGKM operations outstanding"
// (s_waitcnt lgkmcnt(5)) means column K has arrived whatever else the compiler has in flight.  The first column of the NEXT
// product (sPnext) is requested under the last column, so a product never starts by waiting for LDS.
template <int OFF>
__device__ __forceinline__ double m20_lds64(unsigned addr)
{
   double v;
   asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
   return v;
}
template <int K>
__device__ __forceinline__ void m20_acol_asm(unsigned base, double (&A)[5])
{
   A[0] = m20_lds64<(0 * 20 + 4 * K) * 8>(base);
   A[1] = m20_lds64<(4 * 20 + 4 * K) * 8>(base);
   A[2] = m20_lds64<(8 * 20 + 4 * K) * 8>(base);
   A[3] = m20_lds64<(12 * 20 + 4 * K) * 8>(base);
   A[4] = m20_lds64<(16 * 20 + 4 * K) * 8>(base);
}
// wait until at most N LGKM operations are outstanding and tie the five values to the wait, so that nothing using them moves above it
template <int N>
__device__ __forceinline__ void m20_wait(double (&A)[5])
{
   asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]) : "n"(N));
}
__device__ __forceinline__ unsigned m20_lds_addr(const double *p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)p; }
__device__ __forceinline__ void m20_acol(const double *sPn, int aoff, int K, double (&A)[5])
{
#pragma unroll
   for (int I = 0; I < 5; I++) A[I] = sPn[aoff + (4 * I) * 20 + 4 * K];
}
template <int K>
__device__ __forceinline__ void m20_column(const double (&A)[5], const double (&x0)[5], double (&y0)[5], const double (&x1)[5], double (&y1)[5])
{
#pragma unroll
   for (int I = 0; I < 5; I++) {
      y0[I] = M20_MFMA(A[I], x0[K], y0[I]);
      y1[I] = M20_MFMA(A[I], x1[K], y1[I]);
   }
}
__device__ __forceinline__ void m20_matvec2(const double *sPn, const double *sPnext, int aoff, double (&A0)[5], const double (&x0)[5], double (&y0)[5],
                                            const double (&x1)[5], double (&y1)[5])
{
#pragma unroll
   for (int I = 0; I < 5; I++) { y0[I] = 0; y1[I] = 0; }
   const unsigned base = m20_lds_addr(sPn) + aoff * 8, nbase = m20_lds_addr(sPnext) + aoff * 8;
   double A1[5];
   m20_acol_asm<1>(base, A1);  m20_wait<5>(A0);  m20_column<0>(A0, x0, y0, x1, y1);
   m20_acol_asm<2>(base, A0);  m20_wait<5>(A1);  m20_column<1>(A1, x0, y0, x1, y1);
   m20_acol_asm<3>(base, A1);  m20_wait<5>(A0);  m20_column<2>(A0, x0, y0, x1, y1);
   m20_acol_asm<4>(base, A0);  m20_wait<5>(A1);  m20_column<3>(A1, x0, y0, x1, y1);
   m20_acol_asm<0>(nbase, A1); m20_wait<5>(A0);  m20_column<4>(A0, x0, y0, x1, y1);
#pragma unroll
   for (int I = 0; I < 5; I++) A0[I] = A1[I];      // the next product's first column, possibly still in flight: its wait comes first there
}
// ---- hybrid product (jit_generate_m20, default): rows 0-15 of P on v_mfma_f64_16x16x4 (one row block, no padding: its A operand
// is P[lane & 15][4 kb + (lane >> 4)], its accumulator tuple is blocks 0..3 of the partial), rows 16-19 on v_mfma_f64_4x4x4 as
// above.  Same matrix-pipe time (5 x 64 + 5 x 16 cycles per 16 patterns = 25 x 16), but TEN operand fetches per product instead of
// twenty-five: the LDS pipe, which the 4x4x4-only product kept 60 % busy (485 reads per 650 MFMAs with the tip rows, every one
// delivering a 4-fold replicated block), stops being what the MFMAs wait for.  P(t) of a branch in LDS: [kb][lane] in A-operand
// order for the big part (320 doubles), then [kb][k][i] for rows 16-19 (80 doubles).
// Fetch discipline as above (asm ds_read_b64, counted lgkmcnt waits, LDS returns in order): entering a product its five big
// operands are already requested; it requests its five small ones and then the NEXT product's big ones, waits for "all but those
// ten", runs the big MFMAs of both pattern groups, waits for "all but the last five" and runs the small ones in between.
// (As built: the next product's big operands are requested after this product's last big MFMA, into the same registers.)
typedef double m20_v4d __attribute__((ext_vector_type(4)));
#define M20_MFMA16(A, B, C) __builtin_amdgcn_mfma_f64_16x16x4f64((A), (B), (C), 0, 0, 0)
__device__ __forceinline__ void m20h_read_big(unsigned base, int lane, double (&A)[5])      // base: LDS byte address of the node's block
{
   const unsigned a = base + lane * 8;
   A[0] = m20_lds64<0 * 512>(a); A[1] = m20_lds64<1 * 512>(a); A[2] = m20_lds64<2 * 512>(a); A[3] = m20_lds64<3 * 512>(a); A[4] = m20_lds64<4 * 512>(a);
}
__device__ __forceinline__ void m20h_read_small(unsigned base, int lane, double (&A)[5])
{
   const unsigned a = base + 2560 + (((lane >> 4) << 2) + (lane & 3)) * 8;
   A[0] = m20_lds64<0 * 128>(a); A[1] = m20_lds64<1 * 128>(a); A[2] = m20_lds64<2 * 128>(a); A[3] = m20_lds64<3 * 128>(a); A[4] = m20_lds64<4 * 128>(a);
}
__device__ __forceinline__ void m20h_matvec2(const double *sPn, const double *sPnext, int lane, double (&Ab)[5], const double (&x0)[5], double (&y0)[5],
                                             const double (&x1)[5], double (&y1)[5])
{
   double As[5];
   m20h_read_small(m20_lds_addr(sPn), lane, As);
   m20_wait<5>(Ab);
   m20_v4d b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
   double s0 = 0, s1 = 0;
   b0 = M20_MFMA16(Ab[0], x0[0], b0); b1 = M20_MFMA16(Ab[0], x1[0], b1);
   b0 = M20_MFMA16(Ab[1], x0[1], b0); b1 = M20_MFMA16(Ab[1], x1[1], b1);
   m20_wait<0>(As);
   s0 = M20_MFMA(As[0], x0[0], s0); s1 = M20_MFMA(As[0], x1[0], s1);
   b0 = M20_MFMA16(Ab[2], x0[2], b0); b1 = M20_MFMA16(Ab[2], x1[2], b1);
   s0 = M20_MFMA(As[1], x0[1], s0); s1 = M20_MFMA(As[1], x1[1], s1);
   b0 = M20_MFMA16(Ab[3], x0[3], b0); b1 = M20_MFMA16(Ab[3], x1[3], b1);
   s0 = M20_MFMA(As[2], x0[2], s0); s1 = M20_MFMA(As[2], x1[2], s1);
   b0 = M20_MFMA16(Ab[4], x0[4], b0); b1 = M20_MFMA16(Ab[4], x1[4], b1);
   // the big operands have been read by the MFMAs above (sources are read at issue): their registers take the NEXT product's
   // big operands now, which arrive under the remaining small MFMAs and the steps between the products
   m20h_read_big(m20_lds_addr(sPnext), lane, Ab);
   s0 = M20_MFMA(As[3], x0[3], s0); s1 = M20_MFMA(As[3], x1[3], s1);
   s0 = M20_MFMA(As[4], x0[4], s0); s1 = M20_MFMA(As[4], x1[4], s1);
#pragma unroll
   for (int m = 0; m < 4; m++) { y0[m] = b0[m]; y1[m] = b1[m]; }
   y0[4] = s0; y1[4] = s1;
}
// the same product for ONE pattern group (the half units at the end of a workgroup's range, jit_generate_m20)
__device__ __forceinline__ void m20h_matvec1(const double *sPn, const double *sPnext, int lane, double (&Ab)[5], const double (&x0)[5], double (&y0)[5])
{
   double As[5];
   m20h_read_small(m20_lds_addr(sPn), lane, As);
   m20_wait<5>(Ab);
   m20_v4d b0 = {0, 0, 0, 0};
   double s0 = 0;
   b0 = M20_MFMA16(Ab[0], x0[0], b0);
   b0 = M20_MFMA16(Ab[1], x0[1], b0);
   m20_wait<0>(As);
   s0 = M20_MFMA(As[0], x0[0], s0);
   b0 = M20_MFMA16(Ab[2], x0[2], b0);
   s0 = M20_MFMA(As[1], x0[1], s0);
   b0 = M20_MFMA16(Ab[3], x0[3], b0);
   s0 = M20_MFMA(As[2], x0[2], s0);
   b0 = M20_MFMA16(Ab[4], x0[4], b0);
   m20h_read_big(m20_lds_addr(sPnext), lane, Ab);
   s0 = M20_MFMA(As[3], x0[3], s0);
   s0 = M20_MFMA(As[4], x0[4], s0);
#pragma unroll
   for (int m = 0; m < 4; m++) y0[m] = b0[m];
   y0[4] = s0;
}
// ---- trees with more internal branches than LDS holds P(t) blocks for (> 46: more than 49 taxa) --------------------------------------
// The first branches of the walk stay in LDS as above; the others' operands come straight from the operand-order copy pmat_kernel_t<32>
// leaves in global memory (layout 2: [kb][lane] for rows 0-15, then [kb][k][i] for rows 16-19 — the LDS block, 3 200 bytes per branch,
// L2-resident): ten 8-byte loads per lane, requested one product ahead like the LDS ones (the big five into the registers the
// previous product has just read, the small five into a second set), ordinary loads whose waits the compiler counts.
// CG / NG: this / the next product's operands are in global memory (Pc / Pn then point there, else into LDS).
template <bool CG, bool NG>
__device__ __forceinline__ void m20h_matvec2x(const double *Pc, const double *Pn, int lane, double (&Ab)[5], double (&AsN)[5], const double (&x0)[5], double (&y0)[5],
                                              const double (&x1)[5], double (&y1)[5])
{
   double As[5];
   if constexpr (!CG) { m20h_read_small(m20_lds_addr(Pc), lane, As); m20_wait<5>(Ab); }
   else {
#pragma unroll
      for (int i = 0; i < 5; i++) As[i] = AsN[i];
   }
   m20_v4d b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
   double s0 = 0, s1 = 0;
   b0 = M20_MFMA16(Ab[0], x0[0], b0); b1 = M20_MFMA16(Ab[0], x1[0], b1);
   b0 = M20_MFMA16(Ab[1], x0[1], b0); b1 = M20_MFMA16(Ab[1], x1[1], b1);
   if constexpr (!CG) m20_wait<0>(As);
   s0 = M20_MFMA(As[0], x0[0], s0); s1 = M20_MFMA(As[0], x1[0], s1);
   b0 = M20_MFMA16(Ab[2], x0[2], b0); b1 = M20_MFMA16(Ab[2], x1[2], b1);
   s0 = M20_MFMA(As[1], x0[1], s0); s1 = M20_MFMA(As[1], x1[1], s1);
   b0 = M20_MFMA16(Ab[3], x0[3], b0); b1 = M20_MFMA16(Ab[3], x1[3], b1);
   s0 = M20_MFMA(As[2], x0[2], s0); s1 = M20_MFMA(As[2], x1[2], s1);
   b0 = M20_MFMA16(Ab[4], x0[4], b0); b1 = M20_MFMA16(Ab[4], x1[4], b1);
   if constexpr (!NG) m20h_read_big(m20_lds_addr(Pn), lane, Ab);
   else {
#pragma unroll
      for (int i = 0; i < 5; i++) { Ab[i] = Pn[i * 64 + lane]; AsN[i] = Pn[320 + i * 16 + ((lane >> 4) << 2) + (lane & 3)]; }
   }
   s0 = M20_MFMA(As[3], x0[3], s0); s1 = M20_MFMA(As[3], x1[3], s1);
   s0 = M20_MFMA(As[4], x0[4], s0); s1 = M20_MFMA(As[4], x1[4], s1);
#pragma unroll
   for (int m = 0; m < 4; m++) { y0[m] = b0[m]; y1[m] = b1[m]; }
   y0[4] = s0; y1[4] = s1;
}
template <bool CG, bool NG>
__device__ __forceinline__ void m20h_matvec1x(const double *Pc, const double *Pn, int lane, double (&Ab)[5], double (&AsN)[5], const double (&x0)[5], double (&y0)[5])
{
   double As[5];
   if constexpr (!CG) { m20h_read_small(m20_lds_addr(Pc), lane, As); m20_wait<5>(Ab); }
   else {
#pragma unroll
      for (int i = 0; i < 5; i++) As[i] = AsN[i];
   }
   m20_v4d b0 = {0, 0, 0, 0};
   double s0 = 0;
   b0 = M20_MFMA16(Ab[0], x0[0], b0);
   b0 = M20_MFMA16(Ab[1], x0[1], b0);
   if constexpr (!CG) m20_wait<0>(As);
   s0 = M20_MFMA(As[0], x0[0], s0);
   b0 = M20_MFMA16(Ab[2], x0[2], b0);
   s0 = M20_MFMA(As[1], x0[1], s0);
   b0 = M20_MFMA16(Ab[3], x0[3], b0);
   s0 = M20_MFMA(As[2], x0[2], s0);
   b0 = M20_MFMA16(Ab[4], x0[4], b0);
   if constexpr (!NG) m20h_read_big(m20_lds_addr(Pn), lane, Ab);
   else {
#pragma unroll
      for (int i = 0; i < 5; i++) { Ab[i] = Pn[i * 64 + lane]; AsN[i] = Pn[320 + i * 16 + ((lane >> 4) << 2) + (lane & 3)]; }
   }
   s0 = M20_MFMA(As[3], x0[3], s0);
   s0 = M20_MFMA(As[4], x0[4], s0);
#pragma unroll
   for (int m = 0; m < 4; m++) y0[m] = b0[m];
   y0[4] = s0;
}
// tip factors: row `code` of the tip's table, stored [code][st][m] (pmat_kernel layout 2) so that this lane's five states
// 4 m + st are 40 contiguous bytes and the four lanes of a pattern read one 160-byte row
__device__ __forceinline__ void m20_tip(const double *T, int row, int code, int st, double (&v)[5])      // row = doubles per code (20, or 21 in LDS)
{
#ifdef M20_ABL_NOTIP
#pragma unroll
   for (int m = 0; m < 5; m++) v[m] = 0.05 + 0.001 * code;
   return;
#endif
   const double *r = T + code * row + st * 5;
#pragma unroll
   for (int m = 0; m < 5; m++) v[m] = r[m];
}
__device__ __forceinline__ double m20_scale(double (&x)[5])      // NodeScale treesub.c:7200-7230 (maximum over the 20 states: 5 registers x lane bits 4-5)
{
   double mx = 0;
#pragma unroll
   for (int m = 0; m < 5; m++) mx = x[m] > mx ? x[m] : mx;
   double o = __shfl_xor(mx, 16);
   mx = o > mx ? o : mx;
   o = __shfl_xor(mx, 32);
   mx = o > mx ? o : mx;
   if (mx < 1e-300) {
#pragma unroll
      for (int m = 0; m < 5; m++) x[m] = 1.0;
      return -800;
   }
#pragma unroll
   for (int m = 0; m < 5; m++) x[m] /= mx;
   return log(mx);
}
__device__ __forceinline__ void m20_root(const PruneArgs &a, const double (&x)[5], const double (&pis)[5], double lnscale, int iclass, long h, bool own)
{
   double f = 0;
#pragma unroll
   for (int m = 0; m < 5; m++) f = fma(pis[m], x[m], f);
   f += __shfl_xor(f, 16);
   f += __shfl_xor(f, 32);
   if (own) {
      // fx_r treesub.c:7731-7749 / lfun 7782-7798: the floor here, log + scale factors in the reduction kernel (ReduceArgs::raw):
      // a log here runs for the whole wave with 16 useful lanes, on the pipe the MFMAs use
      if (f <= 0) f = (a.mode == PAML_AMD_MODE_LFUN ? 1e-80 : 1e-300);
      a.fhK[(long)iclass * a.n_patt + h] = a.weights[h] > 0 ? f : 0.0;
      if (a.n_scale) a.fscale[(long)iclass * a.n_patt + h] = lnscale;
   }
}

#define JV_PROLOGUE(NS)                                                                                          \
   constexpr int N = NS;                                                                                        \
   const int tid = threadIdx.x;                                                                                 \
   const int tile = blockIdx.x % a.n_tiles, iclass = blockIdx.x / a.n_tiles;                                    \
   const int gene = as_const(a.tiles)[tile].x, h0 = as_const(a.tiles)[tile].y;                                  \
   const int hend = as_const(a.gene_off)[gene + 1];                                                             \
   const int h = h0 + tid;                                                                                      \
   const bool valid = h < hend;                                                                                 \
   const int hc = valid ? h : hend - 1;                                                                         \
   const long pset = (long)gene * a.K + iclass;                                                                 \
   const double *Pint = a.pint + pset * a.n_nodes * (N * N);                                                    \
   const double *Ptip = a.ptip + pset * a.n_nodes * a.tip_words;                                                \
   double lnscale = 0;                                                                                          \
   (void)lnscale;
#define JV_CODE(TIP) ((int)a.z[(long)(TIP)*a.z_stride + hc])
#define JV_ROW(TIP, CODE) (Ptip + (long)(TIP)*a.tip_words + (CODE)*N)

// ---- seamless variant: the operand ring and the tip-code blocks run on across tile boundaries ------------------------
// The workgroup's blocks are numbered through all its tiles; a tile's block J sits in ring buffer (J + roff) & 3 with
// roff advancing by the tile's block count, and blocks J >= NBLK are the next tile's (its P pointers).  Tip codes and
// weight flags of a tile arrive as one small DMA block (PruneArgs::ztiles) in the sZ buffer the previous tile is not
// using.  Nothing at a tile boundary waits on vector memory.
/* JIT_ZB: buffers for the tiles' tip-code blocks.  2 = the next tile's codes arrive while the current tile is walked; 1 = trees
 * whose code block (128 bytes per tip) leaves no room for a second one: the block is replaced between tiles. */
#ifndef JIT_ZB
#define JIT_ZB 2
#endif
/* doubles per tip table: one 32 KB block up to 64 codes; with more (JIT_AMB_OVERFLOW) the table goes on behind its first 64 rows,
 * which are what the ring fetches */
#ifdef JIT_AMB_OVERFLOW
#define JIT_TIPW a.tip_words
#else
#define JIT_TIPW 4096
#endif
#define JIT2_PROLOGUE(ZP)                                                                                        \
   __shared__ __attribute__((aligned(16))) double ring[4 * 4096];                                               \
   __shared__ __attribute__((aligned(16))) unsigned char sZ[JIT_ZB * (ZP)*2048];                                 \
   __shared__ double sPi[4 * 64];                                                                               \
   __shared__ __attribute__((aligned(16))) double sCol[4 * 64 + 32];                                            \
   __shared__ __attribute__((aligned(16))) double sDump[128];                                                   \
   const int tid = threadIdx.x, lane = tid & 63;                                                                \
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                   \
   const int q = lane >> 4, hl = lane & 15;                                                                     \
   const int hw = wave * 16 + hl;                                                                               \
   const int n = a.n;                                                                                           \
   const int total_work = a.n_tiles * a.K;                                                                      \
   int work = blockIdx.x;                                                                                       \
   const double *Pcol = a.pcol, *nPcol = a.pcol;                                                                \
   (void)sCol; (void)sDump; (void)Pcol; (void)nPcol;                                                            \
   int iclass = 0, gene = 0, h0 = 0, hend = 1, h = 0;                                                           \
   int n_tile = 0, n_gene = 0, n_iclass = 0, n_h0 = 0, n_hend = 1;                                              \
   bool valid = false, has_next = false;                                                                        \
   const double *Pint = nullptr, *Ptip = nullptr, *nPint = a.pint, *nPtip = a.ptip;                             \
   double lnscale = 0;                                                                                          \
   int roff = 0, zsel = JIT_ZB - 1;                                                                             \
   (void)hl; (void)n; (void)lnscale; (void)h0;                                                                  \
   if (work >= total_work) return;                                                                              \
   for (int i = tid; i < a.n_pi * 64 && i < 256; i += JIT_WAVES * 64) sPi[i] = a.pi[i];
#define JIT2_NEXT_SET()                                                                                          \
   has_next = work < total_work;                                                                                \
   if (has_next) {                                                                                              \
      n_tile = work % a.n_tiles; n_iclass = work / a.n_tiles;                                                   \
      n_gene = as_const(a.tiles)[n_tile].x; n_h0 = as_const(a.tiles)[n_tile].y;                                 \
      n_hend = as_const(a.gene_off)[n_gene + 1];                                                                \
      nPint = a.pint + ((long)n_gene * a.K + n_iclass) * a.n_nodes * 4096;                                      \
      nPtip = a.ptip + ((long)n_gene * a.K + n_iclass) * a.n_nodes * JIT_TIPW;                                  \
      nPcol = a.pcol + ((long)n_gene * a.K + n_iclass) * a.n_nodes * 64;                                        \
   }   /* past the last tile the n_* values stay: the (unused) prefetches keep reading valid memory */
#define JIT2_ADVANCE(NBLK)                                                                                       \
   iclass = n_iclass; gene = n_gene; h0 = n_h0; hend = n_hend; Pint = nPint; Ptip = nPtip; Pcol = nPcol;        \
   h = h0 + hw; valid = h < hend; lnscale = 0;                                                                  \
   roff = (roff + (NBLK)) & 3; zsel ^= JIT_ZB - 1;
/* the next tile's code block -> the sZ buffer not in use (ZP dword pieces per thread) */
#define JIT2_ISSUE_Z(ZP)                                                                                         \
   {                                                                                                            \
      /* (ZP)*8 pieces of 256 bytes over JIT_WAVES waves; a wave without a piece in the last round still issues one (every  \
       * wave must count the same vector-memory operations) against an empty descriptor, into the dump slot */          \
      _Pragma("unroll") for (int c_ = 0; c_ < ((ZP)*8 + JIT_WAVES - 1) / JIT_WAVES; c_++) {                      \
         const int pi_ = c_ * JIT_WAVES + wave;                                                                 \
         const bool rz_ = pi_ < (ZP)*8;                                                                         \
         dma4(make_rsrc(a.ztiles + (long)n_tile * ((ZP)*2048), rz_ ? (ZP)*2048 : 0),                            \
              rz_ ? (const void *)(sZ + ((zsel ^ 1) & (JIT_ZB - 1)) * ((ZP)*2048) + pi_ * 256) : (const void *)sDump, lane * 4, pi_ * 256);  \
      }                                                                                                         \
   }
/* piece mode (trees of more than 207 tips, jit.h: jit_zplan): a tile's codes are JIT_ZPIECES blocks of (ZP)*2048 bytes, [tile][piece];
 * PIECE of tile TILE -> the one sZ buffer */
#ifndef JIT_ZPIECES
#define JIT_ZPIECES 2
#endif
#define JIT2_ISSUE_ZH(ZP, TILE, PIECE)                                                                            \
   {                                                                                                            \
      _Pragma("unroll") for (int c_ = 0; c_ < ((ZP)*8 + JIT_WAVES - 1) / JIT_WAVES; c_++) {                      \
         const int pi_ = c_ * JIT_WAVES + wave;                                                                 \
         const bool rz_ = pi_ < (ZP)*8;                                                                         \
         dma4(make_rsrc(a.ztiles + ((long)(TILE)*JIT_ZPIECES + (PIECE)) * ((ZP)*2048), rz_ ? (ZP)*2048 : 0),     \
              rz_ ? (const void *)(sZ + pi_ * 256) : (const void *)sDump, lane * 4, pi_ * 256);                  \
      }                                                                                                         \
   }
#define JIT2_BUF(J) (ring + (((J) + roff) & 3) * 4096)
/* the column-60 table that travels with a P block (61 states): 512 bytes, fetched as one dword DMA piece per thread so
 * that every wave issues the same number of vector-memory instructions — waves 0 and 1 carry the data, the descriptor
 * ends after 512 bytes, and the zeros the other waves' lanes read go to a 256-byte dump slot behind the four tables */
#define JIT2_COL(J) (sCol + (((J) + roff) & 3) * 64)
#define JIT2_PIECE_C(SRC, J)                                                                                         \
   dma4(make_rsrc((SRC), 512), wave < 2 ? (const char *)JIT2_COL(J) + wave * 256 : (const char *)(sCol + 256), lane * 4, wave * 256)
#define JIT2_PIECE_PC(J, NODE) JIT2_PIECE_C(Pcol + (long)(NODE)*64, J)
#define JIT2_PIECE_NPC(J, NODE) JIT2_PIECE_C(nPcol + (long)(NODE)*64, J)
/* One DMA piece: 1 KB chunk (C*8 + wave) of block J.  Only the chunks that hold states of the model are fetched — of a P,
 * k-block pairs < JIT_KB2 and row blocks < JIT_RB (chunk = pair * 4 + row block); of a tip table, the first JIT_TCH chunks
 * (two codes each) — the generator emits only the rounds C that contain such a chunk, and inside a round a wave whose
 * chunk is padding still issues its instruction (every wave must count the same vector-memory operations) against an
 * empty descriptor, the zeros going to a dump slot. */
#ifndef JIT_KB2
#define JIT_KB2 8
#define JIT_RB 4
#define JIT_TCH 32
#endif
#define JIT2_PIECE(SRC, J, C, REAL)                                                                                  \
   {                                                                                                                \
      const int ch_ = (C)*JIT_WAVES + wave;                                                                         \
      const bool real_ = ch_ < 32 && (REAL);                                                                                    \
      dma16(make_rsrc((SRC), real_ ? 32768 : 0), real_ ? (const char *)JIT2_BUF(J) + ch_ * 1024 : (const char *)sDump, lane * 16,    \
            ch_ * 1024);                                                                                            \
   }
#define JIT2_REAL_P ((ch_ >> 2) < JIT_KB2 && (ch_ & 3) < JIT_RB)
#define JIT2_REAL_T (ch_ < JIT_TCH)
#define JIT2_PIECE_P(J, NODE, C) JIT2_PIECE(Pint + (long)(NODE)*4096, J, C, JIT2_REAL_P)
#define JIT2_PIECE_T(J, NODE, C) JIT2_PIECE(Ptip + (long)(NODE)*JIT_TIPW, J, C, JIT2_REAL_T)
#define JIT2_PIECE_NP(J, NODE, C) JIT2_PIECE(nPint + (long)(NODE)*4096, J, C, JIT2_REAL_P)
#define JIT2_PIECE_NT(J, NODE, C) JIT2_PIECE(nPtip + (long)(NODE)*JIT_TIPW, J, C, JIT2_REAL_T)
#define JIT2_CODE(ZP, TIP) ((int)sZ[zsel * ((ZP)*2048) + (TIP)*JIT_TP + hw])
#define JIT2_NCODE(ZP, TIP) ((int)sZ[((zsel ^ 1) & (JIT_ZB - 1)) * ((ZP)*2048) + (TIP)*JIT_TP + hw])

}  // namespace paml_amd
