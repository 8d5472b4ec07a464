// eigen_kernels.h — batched eigen-decomposition of reversible rate matrices on the device (gfx950), FP64.
//
// What eigenQREV (tools.c:5023-5110) does on the host for every omega class of every trial point of an optimisation
// (eigenQcodon codeml.c:3229 calls it; 0.3-0.9 ms of one CPU core at n = 61):  Q = S diag(pi) is similar to the symmetric
// A = diag(sqrt pi) Q diag(1 / sqrt pi);  A = R diag(w) R^T;  Root = w (descending), U = diag(1 / sqrt pi) R, V = R^T diag(sqrt pi).
// States of frequency zero are left out of the eigen problem and get Root = 0 and unit rows / columns (tools.c:5040-5105).
//
// One workgroup (EIG_NW waves) per matrix, A and R^T in LDS, cyclic Jacobi in the parallel (round-robin tournament) order: in a round the
// N / 2 disjoint pairs (p, q) are rotated together — a wave works out the angles of its 8 pairs from their 2 x 2 blocks, combines
// their rows of A and R^T (lane = column: conflict-free rows of stride 65) and, after a barrier, their columns of A (lane = row).
// N - 1 rounds visit every pair once (a sweep); 9-10 sweeps bring the off-diagonal part of a 61 x 61 codon matrix below
// 1e-16 ||A|| (quadratic convergence).  Every matrix of a batch has its own workgroup: a gradient's or a line search's several
// hundred decompositions take the time of one, ~0.8 ms, and U, V, Root are written straight into the engine's eigen sets — they never
// cross PCIe.  This is latency-class work (~25 MFLOP per matrix; nothing here wants the matrix cores) and it is bound by the LDS
// pipeline of the one CU a matrix lives on: a round moves 6 KB per pair, 85 us per sweep with 8 waves as with 16 (4 waves: 115 us).
// What an optimiser can save is sweeps: with the warm start (EigenQrevArgs::R0) a matrix that moved by a finite-difference step
// takes 3 sweeps (two that rotate, one that only looks: pairs the stopping rule accepts are branched around), 0.26 ms; after a
// line-search step of 5 % 5 sweeps, 0.43 ms (tools/eigen_probe.py).
#pragma once
#include <hip/hip_runtime.h>

namespace paml_amd {

struct EigenQrevArgs {
   int n;
   const double *Q;          // [n_sets][n * n] rate matrices (row-major; only the lower triangle is read, as eigenQREV does)
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

__global__ __launch_bounds__(EIG_NT) void eigen_qrev_kernel(EigenQrevArgs a)
{
   extern __shared__ double eig_sm[];
   double *sA = eig_sm, *sV = sA + 64 * EIG_LD, *sT = sV + 64 * EIG_LD, *sSp = sT + 64 * EIG_LD, *sW = sSp + 64;
   int *sRank = (int *)(sW + 64);
   __shared__ double sRed[EIG_NW];
   const int n = a.n, N = (n + 1) & ~1, set = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
   const double *Q = a.Q + (size_t)set * n * n, *pi = a.pi + (size_t)set * n;

   if (tid < 64) sSp[tid] = (tid < n && pi[tid] > 1e-100) ? sqrt(pi[tid]) : 0.0;      // 0: the state is left out
   __syncthreads();
   double nrm = 0;
   for (int idx = tid; idx < 64 * 64; idx += EIG_NT) {
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
   for (int w = 0; w < EIG_NW; w++) nrm2 += sRed[w];
   const double thr = 1e-16 * sqrt(nrm2);

   if (const double *r0 = a.R0 ? a.R0[set] : nullptr) {
      // warm start: sV <- R0^T (rows = the previous eigenvectors), sA <- R0^T A R0.  Any orthogonal R0 gives the right answer — the
      // sweeps below run to the same threshold; a good one gives it sooner.  Left-out states: unit rows of R0^T meet zero rows of A,
      // the products are exact zeros and those rows stay out as in the cold start.
      __syncthreads();
      for (int idx = tid; idx < 64 * 64; idx += EIG_NT) {
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
      for (int idx = tid; idx < 64 * 64; idx += EIG_NT) {      // exactly symmetric: the lower triangle from the upper
         const int i = idx >> 6, j = idx & 63;
         if (i > j) sA[i * EIG_LD + j] = sA[j * EIG_LD + i];
      }
      __syncthreads();
   }

   // A wave owns the pairs wv, wv + 4, ... of a round (at most 8) through all three steps, so the rotation angles never leave its
   // registers: lanes 0-7 compute them (from the 2 x 2 blocks the previous round left in LDS), v_readlane hands them to the wave.
   // Two barriers per round: before the rows are combined (the previous round's column pass is complete), and between the row and
   // the column pass.
   int sweep = 0;
   bool converged = false;
   for (; sweep < a.max_sweeps; sweep++) {
      double big = 0;
      for (int r = 0; r < N - 1; r++) {
         double cl = 1, sl = 0;
         int pl = 0, ql = 0;
         {
            const int k = wv + EIG_NW * (lane & (EIG_PW - 1));      // the pair this lane works out (the other lanes repeat the first EIG_PW: no divergence)
            if (k < N / 2) {
               int p = r + k, q = r - k;
               if (p >= N - 1) p -= N - 1;
               if (q < 0) q += N - 1;
               if (k == 0) { p = r; q = N - 1; }
               if (p > q) { const int t = p; p = q; q = t; }
               pl = p; ql = q;
               const double apq = sA[p * EIG_LD + q];
               big = fmax(big, fabs(apq));
               if (fabs(apq) > thr) {      // (an element the stopping rule accepts is left alone: its pair costs no LDS traffic below)
                  // t = tan(phi), the smaller root of t^2 + 2 theta t - 1 = 0 with theta = (aqq - app) / (2 apq), written without theta:
                  // t = 2 apq / (d + sgn(d) sqrt(d^2 + (2 apq)^2)), d = aqq - app.  Reciprocal and reciprocal square root from the
                  // hardware approximations + Newton steps: t only steers the convergence (a step suffices), c = 1 / sqrt(1 + t^2)
                  // must make the rotation orthogonal to the last bit (two steps); s = t c.
                  // (d and a2 are brought to order one first: with app == aqq and |apq| below 1e-154 the squares would underflow to 0,
                  //  the reciprocal square root of 0 is infinite and the Newton step makes a NaN of it — degenerate spectra do this in
                  //  their late sweeps; the scale cancels in t)
                  const double d0 = sA[q * EIG_LD + q] - sA[p * EIG_LD + p], a20 = 2 * apq, sc = 1.0 / fmax(fabs(d0), fabs(a20));
                  const double d = d0 * sc, a2 = a20 * sc, x = d * d + a2 * a2;
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
            }
         }
         double c[EIG_PW], s[EIG_PW], ap[EIG_PW], aq[EIG_PW], vp[EIG_PW], vq[EIG_PW];
         int ip[EIG_PW], iq[EIG_PW], kp[EIG_PW], kq[EIG_PW];
         bool on[EIG_PW];      // wave-uniform (the angles come out of v_readlane): pairs that are not rotated are branched around
         const int lcol = lane < N ? lane : 0;
#pragma unroll
         for (int u = 0; u < EIG_PW; u++) {      // rows p, q of A and of R^T: lane = column; all loads before the first store (disjoint pairs)
            c[u] = eig_readlane(cl, u); s[u] = eig_readlane(sl, u);
            kp[u] = __builtin_amdgcn_readlane(pl, u); kq[u] = __builtin_amdgcn_readlane(ql, u);
            on[u] = s[u] != 0;
            if (on[u]) {
               ip[u] = kp[u] * EIG_LD + lcol; iq[u] = kq[u] * EIG_LD + lcol;
               ap[u] = sA[ip[u]]; aq[u] = sA[iq[u]]; vp[u] = sV[ip[u]]; vq[u] = sV[iq[u]];
            }
         }
#pragma unroll
         for (int u = 0; u < EIG_PW; u++)
            if (on[u] && lane < N) {
               sA[ip[u]] = c[u] * ap[u] - s[u] * aq[u]; sA[iq[u]] = s[u] * ap[u] + c[u] * aq[u];
               sV[ip[u]] = c[u] * vp[u] - s[u] * vq[u]; sV[iq[u]] = s[u] * vp[u] + c[u] * vq[u];
            }
         __syncthreads();
#pragma unroll
         for (int u = 0; u < EIG_PW; u++)      // columns p, q of A: lane = row
            if (on[u]) {
               ip[u] = lcol * EIG_LD + kp[u]; iq[u] = lcol * EIG_LD + kq[u];
               ap[u] = sA[ip[u]]; aq[u] = sA[iq[u]];
            }
#pragma unroll
         for (int u = 0; u < EIG_PW; u++)
            if (on[u] && lane < N) { sA[ip[u]] = c[u] * ap[u] - s[u] * aq[u]; sA[iq[u]] = s[u] * ap[u] + c[u] * aq[u]; }
         __syncthreads();
      }
      // the largest off-diagonal element this sweep met (lanes 0-7 of every wave hold their pairs')
      for (int off = EIG_PW / 2; off; off >>= 1) big = fmax(big, __shfl_xor(big, off));
      if (lane == 0) sRed[wv] = big;
      __syncthreads();
      double bigall = 0;
      for (int w = 0; w < EIG_NW; w++) bigall = fmax(bigall, sRed[w]);
      const bool done = bigall <= thr;
      __syncthreads();
      if (done) { sweep++; converged = true; break; }
   }

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
   for (int idx = tid; idx < n * 64; idx += EIG_NT) {
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
