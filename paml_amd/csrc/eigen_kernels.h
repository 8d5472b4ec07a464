// eigen_kernels.h — batched eigen-decomposition of reversible rate matrices on the device (gfx950), FP64.
//
// What eigenQREV (tools.c:5023-5110) does on the host for every omega class of every trial point of an optimisation
// (eigenQcodon codeml.c:3229 calls it; 0.3-0.9 ms of one CPU core at n = 61):  Q = S diag(pi) is similar to the symmetric
// A = diag(sqrt pi) Q diag(1 / sqrt pi);  A = R diag(w) R^T;  Root = w (descending), U = diag(1 / sqrt pi) R, V = R^T diag(sqrt pi).
// States of frequency zero are left out of the eigen problem and get Root = 0 and unit rows / columns (tools.c:5040-5105).
//
// One workgroup (8 waves) per matrix, A in LDS, cyclic Jacobi in the parallel (round-robin tournament) order: in a round the N / 2
// disjoint pairs (p, q) are rotated together.  Round 6: A <- J^T A J goes by 2 x 2 blocks (a thread applies both rotations of its block
// at once, in place), a block row per half-wave; for the orders 62 (61 sense codons), 60 and 20 one wave keeps R^T in its registers — lane =
// column, the rounds unrolled so that every row is a register pair the compiler knows — and the other orders combine its rows in LDS; the
// next round's blocks are read while its angles are worked out; whether the off-diagonal part is below 1e-16 ||A|| is looked up directly
// after every sweep (8-9 sweeps for a 61 x 61 codon matrix: quadratic convergence) — see the sweeps below.  Every matrix of a batch has
// its own workgroup: a gradient's or a line search's several hundred decompositions take the time of one, and U, V, Root are written
// straight into the engine's eigen sets — they never cross PCIe.  This is latency-class work (~25 MFLOP per matrix; nothing here wants
// the matrix cores): a round is a dependent chain — the (c, s) of the round, the blocks' 16 multiply-adds and their stores, a barrier, the
// angle formula's ~30 dependent operations, a barrier — of ~2 700 cycles, of which the LDS pipeline is busy for a fifth
// (profiles/r06_eigen.txt has the phases by s_memtime).  What an optimiser can save is sweeps: with the warm start
// (EigenQrevArgs::R0) a matrix that moved by a finite-difference step takes 2 sweeps, after a line-search step of 5 % 4
// (tools/eigen_probe.py; times in DESIGN.md section 4 E).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>

namespace paml_amd {

struct EigenQrevArgs {
   int n;
   const double *Q;          // [n_sets][n * n] rate matrices (row-major; only the lower triangle is read, as eigenQREV does) — or, nnz > 0,
                             // [n_sets][nnz] the elements at (rc[2k], rc[2k + 1]), row >= column, everything else zero
   int nnz;
   const int *rc;
   const double *pi;         // [n_sets][n]
   const double *scale;      // [n_sets]: Root = w / scale (the mean rate eigenQcodon divides by)
   double *const *U;         // [n_sets] device pointers of the eigen sets' buffers
   double *const *V;
   double *const *Root;
   int *sweeps;              // [n_sets] sweeps used (diagnostics; null: not wanted)
   int max_sweeps;           // the sweep limit (40; PAML_AMD_EIGEN_SWEEP_LIMIT lowers it: the tests' way to a decomposition that does not converge)
   int *fail;                // pinned host word, set when a set reached the sweep limit (checked by the next synchronous evaluation)
   // warm start (paml_amd_set_eigen_warm_start): R0[set] = the eigenvectors R^T[64][64] (rows, in the order of the roots) the set's
   // previous decomposition left, or null — the Jacobi sweeps then start from R0^T A R0, which is nearly diagonal when the matrix
   // moved a little (a finite-difference step, a line search): 2-4 sweeps instead of 9-10.  Rout[set]: where this one's go (may be R0[set]).
   const double *const *R0;  // [n_sets] or null
   double *const *Rout;      // [n_sets] or null
};

constexpr int EIG_LD = 65;                                           // row stride of the LDS matrices (doubles)
constexpr size_t EIG_LDS_BYTES = (size_t)(3 * 64 * EIG_LD + 64 + 64) * sizeof(double) + 64 * sizeof(int);

__device__ __forceinline__ double eig_readlane(double x, int l)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
   return __hiloint2double(hi, lo);
}

// C[4 x 4 tile of thread] = X Y (TRANS_Y: X Y^T) over 64 x 64 LDS matrices of stride EIG_LD: thread t owns rows 4 (t >> 4) .. + 3,
// columns 4 (t & 15) .. + 3
template <bool TRANS_Y>
__device__ __forceinline__ void eig_mm_tile(const double *X, const double *Y, int tid, double (&c)[4][4])
{
   const int r0 = (tid >> 4) * 4, c0 = (tid & 15) * 4;
#pragma unroll
   for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) c[a][b] = 0;
   for (int k = 0; k < 64; k++) {
      double x[4], y[4];
#pragma unroll
      for (int a = 0; a < 4; a++) x[a] = X[(r0 + a) * EIG_LD + k];
#pragma unroll
      for (int b = 0; b < 4; b++) y[b] = TRANS_Y ? Y[(c0 + b) * EIG_LD + k] : Y[k * EIG_LD + c0 + b];
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
         for (int b = 0; b < 4; b++) c[a][b] = fma(x[a], y[b], c[a][b]);
   }
}

constexpr int EIG_NW = 8;                  // waves per matrix
constexpr int EIG_NT = 64 * EIG_NW;        // threads
constexpr int EIG_PW = 32 / EIG_NW;        // pairs of a round per wave

constexpr __host__ __device__ int eig_pair_p(int N, int r, int k) { return k == 0 ? r : (r + k >= N - 1 ? r + k - (N - 1) : r + k); }
constexpr __host__ __device__ int eig_pair_q(int N, int r, int k) { return k == 0 ? N - 1 : (r - k < 0 ? r - k + N - 1 : r - k); }

// The rounds of the wave that keeps R^T in registers (lane = column, v[row]), unrolled by templates so that the rows of every pair are
// registers the compiler knows; a pair's (c, s) is one LDS read of the same 16 bytes by every lane.  Half of a round's pairs in either
// phase of the round (the barriers are the workgroup's).
template <int NS, int R, int K>
__device__ __forceinline__ void eig_vrot(double (&v)[NS], double c, double sn)
{
   constexpr int p = eig_pair_p(NS, R, K), q = eig_pair_q(NS, R, K);
   const double vp = v[p], vq = v[q];
   v[p] = c * vp - sn * vq;
   v[q] = sn * vp + c * vq;
}
template <int NS, int R, int K0, int... K>
__device__ __forceinline__ void eig_vrots(double (&v)[NS], const double *cs, std::integer_sequence<int, K...>)
{
   // (all the angles of the half round first: 8-byte reads of one address by every lane — as double2, 16-byte reads of one address from
   //  this double-aligned array, they measured ~60 cycles of the LDS pipeline each: profiles/r06_eigen.txt)
   const double c[sizeof...(K)] = {cs[K0 + K]...}, sn[sizeof...(K)] = {cs[32 + K0 + K]...};
   (eig_vrot<NS, R, K0 + K>(v, c[K], sn[K]), ...);
}
template <int NS, int R>
__device__ __forceinline__ void eig_vrounds(double (&v)[NS], const double *sCS, int &par)
{
   if constexpr (R < NS - 1) {
      constexpr int M = NS / 2, MH = (M + 1) / 2;
      const double *cs = sCS + par * 64;
      eig_vrots<NS, R, 0>(v, cs, std::make_integer_sequence<int, MH>{});
      __syncthreads();
      eig_vrots<NS, R, MH>(v, cs, std::make_integer_sequence<int, M - MH>{});
      __syncthreads();
      par ^= 1;
      eig_vrounds<NS, R + 1>(v, sCS, par);
   }
}
template <int NS, int... P>
__device__ __forceinline__ void eig_vload(double (&v)[NS], const double *sV, int lane, std::integer_sequence<int, P...>) { ((v[P] = sV[P * EIG_LD + lane]), ...); }
template <int NS, int... P>
__device__ __forceinline__ void eig_vstore(const double (&v)[NS], double *sV, int lane, std::integer_sequence<int, P...>) { ((sV[P * EIG_LD + lane] = v[P]), ...); }

// NS = the (even) order the register form is compiled for — 62: the 61 sense codons of the universal code, 60: the mitochondrial codes,
// 20: amino acids: wave 0 keeps R^T in its registers, wave 4 (the same SIMD) works out the angles, the other six rotate A; 0: any order,
// every wave rotates A and its share of R^T's rows in LDS, wave 0 works out the angles.
template <int NS>
__global__ __launch_bounds__(EIG_NT) void eigen_qrev_kernel(EigenQrevArgs a)
{
   static_assert(EIG_NW == 8, "the roles of the waves below");
   constexpr int NTH = EIG_NT, NWV = EIG_NW;
   extern __shared__ double eig_sm[];
   double *sA = eig_sm, *sV = sA + 64 * EIG_LD, *sT = sV + 64 * EIG_LD, *sSp = sT + 64 * EIG_LD, *sW = sSp + 64;
   int *sRank = (int *)(sW + 64);
   __shared__ double sRed[EIG_NW + 1];
   const int n = a.n, N = (n + 1) & ~1, set = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   const double *Q = a.Q + (size_t)set * n * n, *pi = a.pi + (size_t)set * n;

   if (tid < 64) sSp[tid] = (tid < n && pi[tid] > 1e-100) ? sqrt(pi[tid]) : 0.0;      // 0: the state is left out
   __syncthreads();
   double nrm = 0;
   if (a.nnz > 0) {      // the sparse hand-over: zeros, then the elements (each lower-triangle element also to its mirror image)
      for (int idx = tid; idx < 64 * 64; idx += NTH) {
         const int i = idx >> 6, j = idx & 63;
         sA[i * EIG_LD + j] = 0.0;
         sV[i * EIG_LD + j] = i == j ? 1.0 : 0.0;
      }
      __syncthreads();
      const double *vals = a.Q + (size_t)set * a.nnz;
      for (int k = tid; k < a.nnz; k += NTH) {
         const int r = a.rc[2 * k], c = a.rc[2 * k + 1];
         const double v = (sSp[r] > 0 && sSp[c] > 0) ? vals[k] * sSp[r] / sSp[c] : 0.0;
         sA[r * EIG_LD + c] = v;
         sA[c * EIG_LD + r] = v;
         nrm += r == c ? v * v : 2 * v * v;
      }
   }
   else
   for (int idx = tid; idx < 64 * 64; idx += NTH) {
      const int i = idx >> 6, j = idx & 63;
      double v = 0;
      if (i < n && j < n) {
         const int r = i > j ? i : j, c = i > j ? j : i;
         if (sSp[r] > 0 && sSp[c] > 0) v = Q[r * n + c] * sSp[r] / sSp[c];
      }
      sA[i * EIG_LD + j] = v;
      sV[i * EIG_LD + j] = i == j ? 1.0 : 0.0;
      nrm += v * v;
   }
   for (int off = 32; off; off >>= 1) nrm += __shfl_xor(nrm, off);
   if (lane == 0) sRed[wv] = nrm;
   __syncthreads();
   double nrm2 = 0;
   for (int w = 0; w < NWV; w++) nrm2 += sRed[w];
   const double thr = 1e-16 * sqrt(nrm2);

   if (const double *r0 = a.R0 ? a.R0[set] : nullptr) {
      // warm start: sV <- R0^T (rows = the previous eigenvectors), sA <- R0^T A R0.  Any orthogonal R0 gives the right answer — the
      // sweeps below run to the same threshold; a good one gives it sooner.  Left-out states: unit rows of R0^T meet zero rows of A,
      // the products are exact zeros and those rows stay out as in the cold start.
      __syncthreads();
      for (int idx = tid; idx < 64 * 64; idx += NTH) {
         const int i = idx >> 6, j = idx & 63;
         sV[i * EIG_LD + j] = (i < n && j < n) ? r0[idx] : (i == j ? 1.0 : 0.0);
      }
      __syncthreads();
      double c[4][4];
      const int tr = (tid >> 4) * 4, tc = (tid & 15) * 4;
      if (tid < 256) {      // (16 x 16 tiles of 4 x 4)
         eig_mm_tile<false>(sV, sA, tid, c);      // T = R0^T A
#pragma unroll
         for (int x = 0; x < 4; x++)
#pragma unroll
            for (int y = 0; y < 4; y++) sT[(tr + x) * EIG_LD + tc + y] = c[x][y];
      }
      __syncthreads();
      if (tid < 256) eig_mm_tile<true>(sT, sV, tid, c);       // B = T R0
      __syncthreads();      // (sA is an operand of the first product)
      if (tid < 256) {
#pragma unroll
         for (int x = 0; x < 4; x++)
#pragma unroll
            for (int y = 0; y < 4; y++) sA[(tr + x) * EIG_LD + tc + y] = c[x][y];
      }
      __syncthreads();
      for (int idx = tid; idx < 64 * 64; idx += NTH) {      // exactly symmetric: the lower triangle from the upper
         const int i = idx >> 6, j = idx & 63;
         if (i > j) sA[i * EIG_LD + j] = sA[j * EIG_LD + i];
      }
      __syncthreads();
   }

   // The sweeps.  A round of the round-robin tournament pairs the N players into N / 2 disjoint (p, q); all of them are rotated together.
   // This is latency-class work on one CU, and what it waits for is instruction issue (a wave64 instruction holds its SIMD for four
   // cycles, the address arithmetic of a round as long as its multiplications) before it is LDS: so
   //   phase 1   A <- J^T A J by 2 x 2 blocks: the thread of block (pair i, pair j) reads its four elements, applies rotation i to the rows
   //             and rotation j to the columns and writes them back (in place: nobody else touches them).  A half-wave is a block row —
   //             lanes = consecutive j: conflict-free rows of stride 65 — and takes 2 or 3 of them; the offsets of its rows and columns
   //             advance from round to round (every player but the last moves up by one) instead of being worked out again.  Rotations
   //             by the identity (pairs the stopping rule accepts) are applied like the others — exact, and nothing waits for a test.
   //             R^T: in the register form wave 0 holds it (lane = column, a register pair per row, the rounds unrolled) and no other wave
   //             of its SIMD has phase-1 work; in the any-order form every wave combines 4 pairs of rows in LDS.
   //   phase 2   the angles of the next round from the diagonal 2 x 2 blocks (lanes 0 .. N / 2 - 1 of one wave, into the other half of sCS).
   // After a sweep the largest off-diagonal element is looked up directly (one pass over A) instead of by a further sweep that only looks.
   double *sCS = sT;      // [2][2][32] c and s of the rounds, double-buffered; sT is free once the warm start is through
   const int m = N / 2;
   auto pair_of = [N](int r, int k, int &p, int &q) {
      p = r + k; q = r - k;
      if (p >= N - 1) p -= N - 1;
      if (q < 0) q += N - 1;
      if (k == 0) { p = r; q = N - 1; }
   };
   auto angles = [&](int r, int par) {      // the lanes < m of one wave
      int p, q;
      pair_of(r, lane, p, q);
      const double apq = sA[(p < q ? p : q) * EIG_LD + (p < q ? q : p)], app = sA[p * EIG_LD + p], aqq = sA[q * EIG_LD + q];      // (the upper triangle steers, as the stopping rule reads it)
      double cl = 1, sl = 0;
      if (fabs(apq) > thr) {      // (an element the stopping rule accepts is left alone)
         // t = tan(phi), the smaller root of t^2 + 2 theta t - 1 = 0 with theta = (aqq - app) / (2 apq), written without theta:
         // t = 2 apq / (d + sgn(d) sqrt(d^2 + (2 apq)^2)), d = aqq - app.  Reciprocal and reciprocal square root from the
         // hardware approximations + Newton steps: t only steers the convergence (a step suffices), c = 1 / sqrt(1 + t^2)
         // must make the rotation orthogonal to the last bit (two steps); s = t c.
         // (d and a2 are brought to order one first: with app == aqq and |apq| below 1e-154 the squares would underflow to 0,
         //  the reciprocal square root of 0 is infinite and the Newton step makes a NaN of it — degenerate spectra do this in
         //  their late sweeps; the scale cancels in t)
         const double d0 = aqq - app, a20 = 2 * apq;
         const int ex = -__builtin_amdgcn_frexp_exp(fmax(fabs(d0), fabs(a20)));      // (a power of two: exact, and no division on the chain)
         const double d = __builtin_amdgcn_ldexp(d0, ex), a2 = __builtin_amdgcn_ldexp(a20, ex), x = d * d + a2 * a2;
         double y = __builtin_amdgcn_rsq(x);
         y = y * (1.5 - 0.5 * x * y * y);
         const double den = d + copysign(x * y, d);
         double rc = __builtin_amdgcn_rcp(den);
         rc = rc * (2.0 - den * rc);
         const double t = a2 * rc, x1 = 1.0 + t * t;
         double c = __builtin_amdgcn_rsq(x1);
         c = c * (1.5 - 0.5 * x1 * c * c);
         c = c * (1.5 - 0.5 * x1 * c * c);
         cl = c; sl = t * c;
      }
      sCS[par * 64 + lane] = cl;
      sCS[par * 64 + 32 + lane] = sl;
   };
   // the roles of the waves, and this thread's blocks of phase 1: block rows ah, ah + AHW, ..., block column hl
   constexpr bool REG = NS > 0;
   constexpr int AHW = REG ? 12 : 16, NR = REG ? 3 : 2, ANGLE_WAVE = REG ? 4 : 0;
   const bool vwave = REG && wv == 0, awave = !REG || (wv != 0 && wv != 4);
   const int ah = REG ? 2 * (wv < 4 ? wv - 1 : wv - 2) + (lane >> 5) : tid >> 5, hl = tid & 31;
   const int kj = hl < m ? hl : 0;
   int ki[NR];
   bool blk[NR];
#pragma unroll
   for (int u = 0; u < NR; u++) {
      blk[u] = awave && hl < m && ah + AHW * u < m;
      ki[u] = blk[u] ? ah + AHW * u : 0;
   }
   const int wrap_c = (N - 1) * 8, wrap_r = (N - 1) * EIG_LD * 8;
   const int vcol = lane < N ? lane : 0;
   double v[REG ? NS : 1];
   if constexpr (REG) {
      __syncthreads();      // (the cold start's identity or the warm start's R0^T is in sV)
      if (vwave) eig_vload<NS>(v, sV, lane, std::make_integer_sequence<int, NS>{});
   }
   int sweep = 0, par = 0;
   bool converged = false;
#ifdef EIG_PROF
   long long prof[5] = {0, 0, 0, 0, 0}, prof2[3] = {0, 0, 0};
   const long long tk0 = clock64(), tw0 = wall_clock64();
#endif
   for (;;) {
      __syncthreads();
      double big = 0;
      for (int idx = tid; idx < 64 * 64; idx += NTH) {
         const int i = idx >> 6, j = idx & 63;
         if (i < j && j < N) big = fmax(big, fabs(sA[i * EIG_LD + j]));
      }
      for (int off = 32; off; off >>= 1) big = fmax(big, __shfl_xor(big, off));
      if (lane == 0) sRed[wv] = big;
      if (sweep == 0 && wv == ANGLE_WAVE && lane < m) angles(0, par);      // (later sweeps: the last round's phase 2 has left them)
      __syncthreads();
      double bigall = 0;
      for (int w = 0; w < NWV; w++) bigall = fmax(bigall, sRed[w]);
      if (bigall <= thr) { converged = true; break; }
      if (sweep >= a.max_sweeps) break;
      if (vwave) {
         if constexpr (REG) eig_vrounds<NS, 0>(v, sCS, par);
      }
      else {
         // byte offsets of this thread's block column (cp, cq) and block rows (rp, rq) in round 0; a round later every player but the
         // last (pair 0's q) has moved up by one, modulo N - 1
         int cp, cq, rp[NR], rq[NR];
         {
            int p, q;
            pair_of(0, kj, p, q);
            cp = p * 8; cq = q * 8;
#pragma unroll
            for (int u = 0; u < NR; u++) {
               pair_of(0, ki[u], p, q);
               rp[u] = p * EIG_LD * 8; rq[u] = q * EIG_LD * 8;
            }
         }
         const int cq_step = kj ? 8 : 0;
         // the blocks of a round are read a phase early — in phase 2 of the round before, as soon as that round's A is complete, beside the
         // angles being worked out (before round 0: here) — so that phase 1 starts with its operands in registers
         int ad[NR][4];
         double b00[NR], b01[NR], b10[NR], b11[NR];
         auto fetch = [&]() {
#pragma unroll
            for (int u = 0; u < NR; u++) {
               ad[u][0] = rp[u] + cp; ad[u][1] = rp[u] + cq; ad[u][2] = rq[u] + cp; ad[u][3] = rq[u] + cq;
               const char *base = (const char *)sA;
               b00[u] = *(const double *)(base + ad[u][0]); b01[u] = *(const double *)(base + ad[u][1]);
               b10[u] = *(const double *)(base + ad[u][2]); b11[u] = *(const double *)(base + ad[u][3]);
            }
         };
         if (awave) fetch();
         for (int r = 0; r < N - 1; r++, par ^= 1) {
#ifdef EIG_PROF
            const long long tp0 = clock64();
#endif
            // ---- phase 1
            const double *cs = sCS + par * 64;
            double vc[EIG_PW], vs[EIG_PW], vp[EIG_PW], vq[EIG_PW];
            int ip[EIG_PW], iq[EIG_PW];
            if constexpr (!REG) {
#pragma unroll
               for (int u = 0; u < EIG_PW; u++) {
                  const int k = wv + EIG_NW * u < m ? wv + EIG_NW * u : 0;      // (a pair that does not exist: pair 0 is loaded, nothing is stored)
                  int p, q;
                  pair_of(r, k, p, q);
                  vc[u] = cs[k]; vs[u] = cs[32 + k];
                  ip[u] = p * EIG_LD + vcol; iq[u] = q * EIG_LD + vcol;
                  vp[u] = sV[ip[u]]; vq[u] = sV[iq[u]];
               }
            }
            if (awave) {
               const double cj = cs[kj], sj = cs[32 + kj];
               double tc[NR], ts[NR];
#pragma unroll
               for (int u = 0; u < NR; u++) { tc[u] = cs[ki[u]]; ts[u] = cs[32 + ki[u]]; }
#ifdef EIG_PROF
               asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
               const long long tpa = clock64();
#endif
#pragma unroll
               for (int u = 0; u < NR; u++)
                  if (blk[u]) {
                     const double ci = tc[u], si = ts[u];
                     const double r00 = ci * b00[u] - si * b10[u], r01 = ci * b01[u] - si * b11[u], r10 = si * b00[u] + ci * b10[u], r11 = si * b01[u] + ci * b11[u];
                     char *wb = (char *)sA;
                     *(double *)(wb + ad[u][0]) = cj * r00 - sj * r01; *(double *)(wb + ad[u][1]) = sj * r00 + cj * r01;
                     *(double *)(wb + ad[u][2]) = cj * r10 - sj * r11; *(double *)(wb + ad[u][3]) = sj * r10 + cj * r11;
                  }
#ifdef EIG_PROF
               const long long tpb = clock64();
               asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
               const long long tpc = clock64();
               prof2[0] += tpa - tp0; prof2[1] += tpb - tpa; prof2[2] += tpc - tpb;
#endif
               // the next round's offsets
               cp += 8; cp = cp == wrap_c ? 0 : cp;
               cq += cq_step; cq = (cq == wrap_c && kj) ? 0 : cq;
#pragma unroll
               for (int u = 0; u < NR; u++) {
                  rp[u] += EIG_LD * 8; rp[u] = rp[u] == wrap_r ? 0 : rp[u];
                  if (ki[u]) { rq[u] += EIG_LD * 8; rq[u] = rq[u] == wrap_r ? 0 : rq[u]; }
               }
            }
            if constexpr (!REG) {
#pragma unroll
               for (int u = 0; u < EIG_PW; u++)
                  if (wv + EIG_NW * u < m && lane < N) { sV[ip[u]] = vc[u] * vp[u] - vs[u] * vq[u]; sV[iq[u]] = vs[u] * vp[u] + vc[u] * vq[u]; }
            }
#ifdef EIG_PROF
            const long long tp1 = clock64();
#endif
            __syncthreads();
#ifdef EIG_PROF
            const long long tp2 = clock64();
#endif
            // ---- phase 2: the next round's angles (after the sweep's last round: round 0 of the next sweep)
            if (wv == ANGLE_WAVE && lane < m) angles(r + 1 < N - 1 ? r + 1 : 0, par ^ 1);
            if (awave) fetch();
#ifdef EIG_PROF
            const long long tp3 = clock64();
#endif
            __syncthreads();
#ifdef EIG_PROF
            prof[0] += tp1 - tp0; prof[1] += tp2 - tp1; prof[2] += tp3 - tp2; prof[3] += clock64() - tp3; prof[4]++;
#endif
         }
      }
      sweep++;
   }
   if constexpr (REG) {
      if (vwave) eig_vstore<NS>(v, sV, lane, std::make_integer_sequence<int, NS>{});
   }

#ifdef EIG_PROF
   if (set == 0 && (tid == 64 || tid == 64 * 4 || tid == 64 * 7))
      printf("eig prof wave %d: rounds %lld  phase1 %lld  bar1 %lld  phase2 %lld  bar2 %lld cycles per round; loads %lld compute+stores issued %lld stores done %lld; sweeps part %lld cycles = %lld ticks of 100 MHz\n", wv, prof[4],
             prof[0] / prof[4], prof[1] / prof[4], prof[2] / prof[4], prof[3] / prof[4], prof2[0] / prof[4], prof2[1] / prof[4], prof2[2] / prof[4], clock64() - tk0, wall_clock64() - tw0);
#endif
   // roots descending (ties: by position), then U = R / sqrt(pi), V = R^T sqrt(pi); left-out states: unit rows / columns, Root = 0
   if (tid < 64) sW[tid] = tid < n ? sA[tid * EIG_LD + tid] : 0.0;
   __syncthreads();
   if (tid < n) {
      const double w = sW[tid];
      int rk = 0;
      for (int j = 0; j < n; j++) rk += (sW[j] > w || (sW[j] == w && j < tid)) ? 1 : 0;
      sRank[tid] = rk;
      a.Root[set][rk] = w / a.scale[set];
   }
   __syncthreads();
   double *U = a.U[set], *V = a.V[set], *Rout = a.Rout ? a.Rout[set] : nullptr;
   for (int idx = tid; idx < n * 64; idx += NTH) {
      const int p = idx >> 6, i = idx & 63;      // eigenvector p = row p of R^T
      const int rk = sRank[p];
      if (Rout) Rout[rk * 64 + i] = i < n ? sV[p * EIG_LD + i] : 0.0;
      if (i >= n) continue;
      const double sp = sSp[i] > 0 ? sSp[i] : 1.0, v = sV[p * EIG_LD + i];
      V[rk * n + i] = v * sp;
      U[i * n + rk] = v / sp;
   }
   if (a.sweeps && tid == 0) a.sweeps[set] = converged ? sweep : -1;      // (-1: the sweep limit was reached: paml_amd_eigen_counters shows it)
   if (!converged && a.fail && tid == 0) __hip_atomic_store(a.fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // ... and the next evaluation returns PAML_AMD_ENOCONV
}

}  // namespace paml_amd
