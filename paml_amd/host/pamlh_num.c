/* pamlh_num.c — numerics of the host-side model layer (plain C, written fresh):
 * symmetric eigen-solver, reversible-Q decomposition, discrete gamma / beta classes.
 * Behaviour follows the reference's tools.c: eigenQREV 5023-5110 (sqrt(pi) symmetrisation, roots sorted
 * descending, U = R/sqrt(pi), V = R^T sqrt(pi)), DiscreteGamma 2601-2627 (mean of category),
 * DiscreteNSsites codeml.c:2846 (median quantiles of beta for M7/M8).  Only the results have to agree with the
 * reference (P(t) is independent of eigenvector signs), so the algorithms are chosen for robustness:
 * cyclic Jacobi for the eigen problem, series / continued fractions + safeguarded Newton for the quantiles. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pamlh_internal.h"

/* Symmetric eigenproblem: A (n x n symmetric, destroyed) -> eigenvalues w[n], eigenvectors R[i*n+k] (column k).
 * Householder reduction to tridiagonal form, then QL iterations with implicit Wilkinson shifts on the tridiagonal
 * matrix, accumulating the rotations into R (O(n^3) with a small constant: ~0.3 ms at n = 61, which matters because
 * every NSsites class of every trial point of an optimisation needs one).  Falls back to cyclic Jacobi if the QL
 * iteration does not settle. */
static void jacobi_sym(double *A, int n, double *w, double *R)
{
   int i, j, k, sweep;
   for (i = 0; i < n; i++)
      for (j = 0; j < n; j++) R[i * n + j] = (i == j);
   for (sweep = 0; sweep < 100; sweep++) {
      double off = 0;
      for (i = 0; i < n; i++)
         for (j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
      if (off < 1e-300) break;
      for (i = 0; i < n - 1; i++)
         for (j = i + 1; j < n; j++) {
            double apq = A[i * n + j], app, aqq, theta, t, c, s;
            if (fabs(apq) < 1e-300) continue;
            app = A[i * n + i];
            aqq = A[j * n + j];
            theta = (aqq - app) / (2 * apq);
            t = (theta >= 0 ? 1 : -1) / (fabs(theta) + sqrt(theta * theta + 1));
            c = 1 / sqrt(t * t + 1);
            s = t * c;
            for (k = 0; k < n; k++) {
               double akp = A[k * n + i], akq = A[k * n + j];
               A[k * n + i] = c * akp - s * akq;
               A[k * n + j] = s * akp + c * akq;
            }
            for (k = 0; k < n; k++) {
               double apk = A[i * n + k], aqk = A[j * n + k];
               A[i * n + k] = c * apk - s * aqk;
               A[j * n + k] = s * apk + c * aqk;
            }
            for (k = 0; k < n; k++) {
               double rkp = R[k * n + i], rkq = R[k * n + j];
               R[k * n + i] = c * rkp - s * rkq;
               R[k * n + j] = s * rkp + c * rkq;
            }
         }
   }
   for (i = 0; i < n; i++) w[i] = A[i * n + i];
}

void pamlh_eigen_sym(double *A, int n, double *w, double *R)
{
   double *dg = w, *e = (double *)malloc(n * sizeof(double)), *v = (double *)malloc(n * sizeof(double));
   double *pv = (double *)malloc(n * sizeof(double)), *A0 = (double *)malloc((size_t)n * n * sizeof(double));
   int i, j, k, l, m, ok = 1;
   memcpy(A0, A, (size_t)n * n * sizeof(double));
   /* R accumulates Q = H_0 H_1 ... with A = Q T Q^T */
   for (i = 0; i < n; i++)
      for (j = 0; j < n; j++) R[i * n + j] = (i == j);
   for (k = 0; k < n - 2; k++) {
      /* reflector that zeroes column k below the subdiagonal */
      double alpha, norm2 = 0, vnorm2, beta, kfac = 0;
      for (i = k + 1; i < n; i++) norm2 += A[i * n + k] * A[i * n + k];
      if (norm2 == 0) continue;
      alpha = A[(k + 1) * n + k] > 0 ? -sqrt(norm2) : sqrt(norm2);
      for (i = 0; i < n; i++) v[i] = 0;
      v[k + 1] = A[(k + 1) * n + k] - alpha;
      for (i = k + 2; i < n; i++) v[i] = A[i * n + k];
      vnorm2 = norm2 - A[(k + 1) * n + k] * A[(k + 1) * n + k] + v[k + 1] * v[k + 1];
      if (vnorm2 == 0) continue;
      beta = 2 / vnorm2;
      /* A <- H A H with H = I - beta v v^T:  p = beta A v,  q = p - (beta/2)(p.v) v,  A -= v q^T + q v^T */
      /* (rows and columns < k are already tridiagonal and v is zero up to k: only the trailing block changes) */
      for (i = k; i < n; i++) {
         double t = 0;
         for (j = k + 1; j < n; j++) t += A[i * n + j] * v[j];
         pv[i] = beta * t;
      }
      for (i = k + 1; i < n; i++) kfac += pv[i] * v[i];
      kfac *= beta / 2;
      for (i = k; i < n; i++) pv[i] -= kfac * v[i];
      for (i = k; i < n; i++)
         for (j = k; j < n; j++) A[i * n + j] -= v[i] * pv[j] + pv[i] * v[j];
      /* R <- R H */
      for (i = 0; i < n; i++) {
         double t = 0;
         for (j = k + 1; j < n; j++) t += R[i * n + j] * v[j];
         t *= beta;
         for (j = k + 1; j < n; j++) R[i * n + j] -= t * v[j];
      }
   }
   for (i = 0; i < n; i++) dg[i] = A[i * n + i];
   for (i = 0; i + 1 < n; i++) e[i] = A[(i + 1) * n + i];
   if (n > 0) e[n - 1] = 0;
   /* QL with implicit shifts on (dg, e); rotations act on pairs of eigenvector columns: work on the transpose so that
    * they are contiguous rows */
   for (i = 0; i < n; i++)
      for (j = i + 1; j < n; j++) { const double t = R[i * n + j]; R[i * n + j] = R[j * n + i]; R[j * n + i] = t; }
   for (l = 0; l < n && ok; l++) {
      int iter = 0;
      for (;;) {
         for (m = l; m + 1 < n; m++) {
            const double dd = fabs(dg[m]) + fabs(dg[m + 1]);
            if (fabs(e[m]) <= 2.3e-16 * dd) break;
         }
         if (m == l) break;
         if (++iter > 60) { ok = 0; break; }
         {
            double g = (dg[l + 1] - dg[l]) / (2 * e[l]), r = hypot(g, 1.0), s = 1, c = 1, pp = 0;
            g = dg[m] - dg[l] + e[l] / (g + (g >= 0 ? fabs(r) : -fabs(r)));
            for (i = m - 1; i >= l; i--) {
               double f = s * e[i], bb = c * e[i];
               r = hypot(f, g);
               e[i + 1] = r;
               if (r == 0) { dg[i + 1] -= pp; e[m] = 0; break; }
               s = f / r;
               c = g / r;
               g = dg[i + 1] - pp;
               r = (dg[i] - g) * s + 2 * c * bb;
               pp = s * r;
               dg[i + 1] = g + pp;
               g = c * r - bb;
               {
                  double *r0 = R + (size_t)i * n, *r1 = r0 + n;
                  for (k = 0; k < n; k++) {
                     const double t = r1[k];
                     r1[k] = s * r0[k] + c * t;
                     r0[k] = c * r0[k] - s * t;
                  }
               }
            }
            if (r == 0 && i >= l) continue;
            dg[l] -= pp;
            e[l] = g;
            e[m] = 0;
         }
      }
   }
   for (i = 0; i < n; i++)
      for (j = i + 1; j < n; j++) { const double t = R[i * n + j]; R[i * n + j] = R[j * n + i]; R[j * n + i] = t; }
   if (!ok) jacobi_sym(A0, n, w, R);
   free(e); free(v); free(pv); free(A0);
}

/* Reversible Q = S diag(pi): Root (descending), U, V with Q = U diag(Root) V  (tools.c:5023-5110).
 * States with pi == 0 are not expected here (the callers use strictly positive frequencies). */
static void eigen_qrev_positive(const double *Q, const double *pi, int n, double *Root, double *U, double *V)
{
   double *A = (double *)malloc((size_t)n * n * sizeof(double)), *R = (double *)malloc((size_t)n * n * sizeof(double));
   double *w = (double *)malloc(n * sizeof(double)), *sp = (double *)malloc(n * sizeof(double));
   int *ord = (int *)malloc(n * sizeof(int)), i, j, k;
   for (i = 0; i < n; i++) sp[i] = sqrt(pi[i]);
   for (i = 0; i < n; i++)
      for (j = 0; j <= i; j++) A[i * n + j] = A[j * n + i] = Q[i * n + j] * sp[i] / sp[j];
   pamlh_eigen_sym(A, n, w, R);
   for (i = 0; i < n; i++) ord[i] = i;
   for (i = 1; i < n; i++) {           /* insertion sort, descending */
      int o = ord[i];
      for (j = i; j > 0 && w[ord[j - 1]] < w[o]; j--) ord[j] = ord[j - 1];
      ord[j] = o;
   }
   for (k = 0; k < n; k++) {
      Root[k] = w[ord[k]];
      for (i = 0; i < n; i++) {
         U[i * n + k] = R[i * n + ord[k]] / sp[i];
         V[k * n + i] = R[i * n + ord[k]] * sp[i];
      }
   }
   free(A); free(R); free(w); free(sp); free(ord);
}

/* Q = S diag(pi) = U diag(Root) V.  States of frequency zero (an observed codon table of a small data set) are left out of the
 * eigen problem and get Root = 0 and unit rows / columns of U and V — the chain never enters or leaves them (eigenQREV tools.c:5040-5105). */
void pamlh_eigen_qrev(const double *Q, const double *pi, int n, double *Root, double *U, double *V)
{
   int *idx = (int *)malloc(n * sizeof(int)), m = 0, i, j;
   for (i = 0; i < n; i++) if (pi[i] > 1e-100) idx[m++] = i;
   if (m == n) eigen_qrev_positive(Q, pi, n, Root, U, V);
   else {
      double *Qr = (double *)calloc((size_t)m * m + 1, sizeof(double)), *pr = (double *)calloc((size_t)m + 1, sizeof(double));
      double *Rr = (double *)malloc(m * sizeof(double)), *Ur = (double *)malloc((size_t)m * m * sizeof(double)), *Vr = (double *)malloc((size_t)m * m * sizeof(double));
      for (i = 0; i < m; i++) { pr[i] = pi[idx[i]]; for (j = 0; j < m; j++) Qr[i * m + j] = Q[idx[i] * n + idx[j]]; }
      eigen_qrev_positive(Qr, pr, m, Rr, Ur, Vr);
      for (i = 0; i < n; i++) { Root[i] = 0; for (j = 0; j < n; j++) U[i * n + j] = V[i * n + j] = (i == j); }
      for (i = 0; i < m; i++) {
         Root[idx[i]] = Rr[i];
         for (j = 0; j < m; j++) { U[idx[i] * n + idx[j]] = Ur[i * m + j]; V[idx[i] * n + idx[j]] = Vr[i * m + j]; }
      }
      free(Qr); free(pr); free(Rr); free(Ur); free(Vr);
   }
   free(idx);
}

/* regularised lower incomplete gamma P(a, x) */
double pamlh_gammp(double a, double x)
{
   const double lga = lgamma(a);
   int i;
   if (x <= 0) return 0;
   if (x < a + 1) {                       /* series */
      double ap = a, sum = 1 / a, del = sum;
      for (i = 0; i < 2000; i++) {
         ap += 1; del *= x / ap; sum += del;
         if (fabs(del) < fabs(sum) * 1e-17) break;
      }
      return sum * exp(-x + a * log(x) - lga);
   }
   else {                                  /* continued fraction (modified Lentz) for Q(a,x) */
      double b = x + 1 - a, c = 1 / 1e-300, d = 1 / b, h = d;
      for (i = 1; i < 2000; i++) {
         double an = -i * (i - a), del;
         b += 2;
         d = an * d + b; if (fabs(d) < 1e-300) d = 1e-300;
         c = b + an / c; if (fabs(c) < 1e-300) c = 1e-300;
         d = 1 / d; del = d * c; h *= del;
         if (fabs(del - 1) < 1e-16) break;
      }
      return 1 - exp(-x + a * log(x) - lga) * h;
   }
}

/* quantile of G(alpha, beta): x with P(alpha, beta x) = p */
double pamlh_quantile_gamma(double p, double alpha, double beta)
{
   double lo = 0, hi = alpha + 10 * sqrt(alpha) + 10, x;
   int i;
   while (pamlh_gammp(alpha, hi) < p) hi *= 2;
   x = 0.5 * (lo + hi);
   for (i = 0; i < 200; i++) {
      double f = pamlh_gammp(alpha, x) - p;
      double dens = exp(-x + (alpha - 1) * log(x) - lgamma(alpha)), xn;
      if (f > 0) hi = x; else lo = x;
      xn = dens > 0 ? x - f / dens : -1;
      if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
      if (fabs(xn - x) <= 1e-15 * fabs(x) + 1e-300) { x = xn; break; }
      x = xn;
   }
   return x / beta;
}

/* DiscreteGamma, mean of each of K equal-probability categories (tools.c:2601-2627), beta = alpha */
void pamlh_discrete_gamma(double *freqK, double *rK, double alpha, int K)
{
   int i;
   double prev = 0;
   for (i = 0; i < K; i++) {
      double cum = 1;
      if (i < K - 1) cum = pamlh_gammp(alpha + 1, pamlh_quantile_gamma((i + 1.0) / K, alpha, alpha) * alpha);
      rK[i] = (cum - prev) * K;            /* mean alpha/beta = 1 */
      prev = cum;
      freqK[i] = 1.0 / K;
   }
}

/* regularised incomplete beta I_x(a, b): continued fraction (Lentz) */
static double betacf(double a, double b, double x)
{
   double qab = a + b, qap = a + 1, qam = a - 1, c = 1, d = 1 - qab * x / qap, h;
   int m;
   if (fabs(d) < 1e-300) d = 1e-300;
   d = 1 / d; h = d;
   for (m = 1; m < 5000; m++) {
      int m2 = 2 * m;
      double aa = m * (b - m) * x / ((qam + m2) * (a + m2)), del;
      d = 1 + aa * d; if (fabs(d) < 1e-300) d = 1e-300;
      c = 1 + aa / c; if (fabs(c) < 1e-300) c = 1e-300;
      d = 1 / d; h *= d * c;
      aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
      d = 1 + aa * d; if (fabs(d) < 1e-300) d = 1e-300;
      c = 1 + aa / c; if (fabs(c) < 1e-300) c = 1e-300;
      d = 1 / d; del = d * c; h *= del;
      if (fabs(del - 1) < 1e-16) break;
   }
   return h;
}

/* I_x(a, b) given lb = log B(a, b) (the three lgamma calls are the caller's, once per (a, b)) */
static double betai_lb(double a, double b, double x, double lb)
{
   double bt;
   if (x <= 0) return 0;
   if (x >= 1) return 1;
   bt = exp(a * log(x) + b * log1p(-x) - lb);
   if (x < (a + 1) / (a + b + 2)) return bt * betacf(a, b, x) / a;
   return 1 - bt * betacf(b, a, 1 - x) / b;
}

double pamlh_betai(double a, double b, double x) { return betai_lb(a, b, x, lgamma(a) + lgamma(b) - lgamma(a + b)); }

/* x with I_x(p, q) = prob.  A bracket [lo, hi] that every evaluation shrinks, and inside it Newton steps — on log I as a function
 * of log x below the median (M7 / M8 classes can sit at 1e-30, where I ~ x^p is a power law and that iteration is nearly exact),
 * on I as a function of x above it; a step that leaves the bracket is replaced by a bisection (geometric while the bracket spans
 * more than a factor of 4).  6-10 evaluations of I instead of the ~60 of plain bisection: DiscreteNSsites (codeml.c:2846) is called
 * for every trial point of an M7 / M8 optimisation, ten quantiles each. */
double pamlh_quantile_beta(double prob, double p, double q)
{
   const double lb = lgamma(p) + lgamma(q) - lgamma(p + q);
   double lo = 0, hi = 1, x = p / (p + q);
   int i;
   if (prob <= 0) return 0;
   if (prob >= 1) return 1;
   for (i = 0; i < 400; i++) {
      const double I = betai_lb(p, q, x, lb), f = I - prob;
      double xn = -1;
      if (f > 0) hi = x; else lo = x;
      if (hi - lo <= 1e-16 * hi || f == 0) break;
      if (lo == 0 && hi < 1e-300) break;
      if (I > 0 && I < 1) {
         const double ld = (p - 1) * log(x) + (q - 1) * log1p(-x) - lb;      /* log of the density at x */
         if (prob < 0.5) xn = x * exp(-log(I / prob) * I / exp(ld + log(x)));
         else xn = x - f / exp(ld);
      }
      if (!(xn > lo && xn < hi)) xn = (lo > 0 && hi / lo > 4) ? sqrt(lo * hi) : (lo == 0 && i > 8) ? hi * 1e-3 : 0.5 * (lo + hi);
      if (fabs(xn - x) <= 1e-16 * x) { x = xn; break; }
      x = xn;
   }
   return x;
}

/* ------------------------------------------------------------------ auto-discrete-gamma (AutodGamma tools.c:2630)
 * The K x K transition matrix between the rate classes of neighbouring sites: the K equal-probability classes are cut on a
 * standard normal scale, and M[i][j] = K * Pr(class i at one site, class j at the next) under a bivariate normal with
 * correlation rho.  The reference builds it from the published approximations restated here — Odeh & Evans (1974, AS 70) for
 * the normal quantile, Adams (1969) / Hill (1973, AS 66) for the normal integral, and Genz (2004) for the bivariate normal
 * tail L(h, k, r) with 16 / 32-point Gauss-Legendre rules — so that the same rho gives the same M to rounding. */
double pamlh_quantile_normal(double prob)
{
   static const double a[5] = {-.322232431088, -1, -.342242088547, -.0204231210245, -.453642210148e-4};
   static const double b[5] = {.0993484626060, .588581570495, .531103462366, .103537752850, .0038560700634};
   const double p1 = prob < 0.5 ? prob : 1 - prob;
   double y, z;
   if (p1 < 1e-20) z = 999;
   else {
      y = sqrt(log(1 / (p1 * p1)));
      z = y + ((((y * a[4] + a[3]) * y + a[2]) * y + a[1]) * y + a[0]) / ((((y * b[4] + b[3]) * y + b[2]) * y + b[1]) * y + b[0]);
   }
   return prob < 0.5 ? -z : z;
}

double pamlh_cdf_normal(double x)
{
   const double ax = fabs(x), y = x * x / 2;
   double p;
   if (ax < 1.28)
      p = .5 - ax * (.398942280444 - .399903438504 * y / (y + 5.75885480458 - 29.8213557808 / (y + 2.62433121679 + 48.6959930692 / (y + 5.92885724438))));
   else
      p = 0.398942280385 * exp(-y) /
          (ax - 3.8052e-8 + 1.00000615302 / (ax + 3.98064794e-4 + 1.98615381364 / (ax - 0.151679116635 + 5.29330324926 /
          (ax + 4.8385912808 - 15.1508972451 / (ax + 0.742380924027 + 30.789933034 / (ax + 3.99019417011))))));
   return x < 0 ? p : 1 - p;
}

/* positive nodes and weights of the n-point Gauss-Legendre rule (n even): Newton on P_n */
static void gauss_legendre_half(int n, double *x, double *w)
{
   int i, k, it;
   for (i = 0; i < n / 2; i++) {
      double z = cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 1;
      for (it = 0; it < 100; it++) {
         double p1 = 1, p2 = 0, p3, dz;
         for (k = 1; k <= n; k++) { p3 = p2; p2 = p1; p1 = ((2.0 * k - 1) * z * p2 - (k - 1.0) * p3) / k; }
         pp = n * (z * p1 - p2) / (z * z - 1);
         dz = p1 / pp;
         z -= dz;
         if (fabs(dz) < 1e-16) break;
      }
      x[i] = z;
      w[i] = 2 / ((1 - z * z) * pp * pp);
   }
}

/* Bivariate normal upper-tail probability L(lo, up, r) = Pr(X > lo, Y > up), lo <= up, after Genz (2004).
 * Moderate correlation: the single-integral form over the arcsine of the correlation (his equation 3).  High correlation
 * (|r| >= 0.925): the expansion around |r| = 1 — an analytic part plus a Gauss-Legendre integral of the remainder (equation 6). */
static double bvn_tail_moderate(double lo, double up, double r, int n, const double *node, const double *wt)
{
   double acc = 0;
   if (fabs(r) > 1e-10) {
      const double half_asin = asin(r) / 2, q = (lo * lo + up * up) / 2;
      int i, side;
      for (i = 0; i < n / 2; i++)
         for (side = -1; side <= 1; side += 2) {
            const double sn = sin(half_asin * (1 + side * node[i]));
            acc += wt[i] * exp((sn * lo * up - q) / (1 - sn * sn));
         }
      acc *= half_asin / (2 * M_PI);
   }
   return acc + pamlh_cdf_normal(-lo) * pamlh_cdf_normal(-up);
}

static double bvn_tail_high(double lo, double up, double r, int n, const double *node, const double *wt)
{
   const double up_s = r >= 0 ? up : -up, prod = r >= 0 ? lo * up : -lo * up;      /* reflected for negative correlation */
   double tail = 0;
   if (fabs(r) < 1) {
      const double one_m_r2 = 1 - r * r, root = sqrt(one_m_r2), gap = fabs(lo - up_s), gap2 = gap * gap;
      const double c1 = (4 - prod) / 8, c2 = (12 - prod) / 16;
      double ex = -(gap2 / one_m_r2 + prod) / 2;
      int i, side;
      if (ex > -500) tail = root * exp(ex) * (1 - c1 * (gap2 - one_m_r2) * (1 - c2 * gap2 / 5) / 3 + c1 * c2 * one_m_r2 * one_m_r2 / 5);
      if (prod > -500) tail -= exp(-prod / 2) * sqrt(2 * M_PI) * pamlh_cdf_normal(-gap / root) * gap * (1 - c1 * gap2 * (1 - c2 * gap2 / 5) / 3);
      for (i = 0; i < n / 2; i++)
         for (side = -1; side <= 1; side += 2) {
            const double u = root / 2 * (1 + side * node[i]), u2 = u * u, rs = sqrt(1 - u2);
            ex = -(gap2 / u2 + prod) / 2;
            if (ex > -500) tail += root / 2 * wt[i] * exp(ex) * (exp(-prod * (1 - rs) / (2 * (1 + rs))) / rs - (1 + c1 * u2 * (1 + c2 * u2)));
         }
      tail /= -2 * M_PI;
   }
   if (r > 0) return tail + pamlh_cdf_normal(-(lo > up ? lo : up));
   tail = -tail;
   if (r < 0 && lo + up < 0) tail += pamlh_cdf_normal(-lo) - pamlh_cdf_normal(up);
   return tail;
}

double pamlh_lbinormal(double h, double k, double r)
{
   const int n = fabs(r) < 0.3 ? 16 : 32;      /* the rule sizes the reference uses */
   const double lo = h < k ? h : k, up = h < k ? k : h;
   double node[16], wt[16], v;
   gauss_legendre_half(n, node, wt);
   v = fabs(r) < 0.925 ? bvn_tail_moderate(lo, up, r, n, node, wt) : bvn_tail_high(lo, up, r, n, node, wt);
   return v < 0 ? 0 : v;
}

void pamlh_autod_gamma(double *M, double *freqK, double *rK, double alpha, double rho, int K)
{
   double pt[64];
   int i, j, s;
   for (i = 0; i < K - 1; i++) pt[i] = pamlh_quantile_normal((i + 1.0) / K);
   for (i = 0; i < K; i++)
      for (j = 0; j < K; j++) M[i * K + j] = pamlh_lbinormal(-(i < K - 1 ? pt[i] : 20), -(j < K - 1 ? pt[j] : 20), rho);      /* cumulative */
   for (s = 2 * (K - 1); s >= 0; s--)         /* cumulative -> cell probability x K, from the far corner inwards */
      for (i = 0; i < K; i++) {
         double y = 0;
         j = s - i;
         if (j < 0 || j >= K) continue;
         if (i > 0) y -= M[(i - 1) * K + j];
         if (j > 0) y -= M[i * K + j - 1];
         if (i > 0 && j > 0) y += M[(i - 1) * K + j - 1];
         M[i * K + j] = (M[i * K + j] + y) * K;
      }
   pamlh_discrete_gamma(freqK, rK, alpha, K);
}
