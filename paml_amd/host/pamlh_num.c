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

/* Cyclic Jacobi: A (n x n symmetric, destroyed) -> eigenvalues w[n], eigenvectors R[i*n+k] (column k). */
void pamlh_eigen_sym(double *A, int n, double *w, double *R)
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

/* Reversible Q = S diag(pi): Root (descending), U, V with Q = U diag(Root) V  (tools.c:5023-5110).
 * States with pi == 0 are not expected here (the callers use strictly positive frequencies). */
void pamlh_eigen_qrev(const double *Q, const double *pi, int n, double *Root, double *U, double *V)
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

double pamlh_betai(double a, double b, double x)
{
   double bt;
   if (x <= 0) return 0;
   if (x >= 1) return 1;
   bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x));
   if (x < (a + 1) / (a + b + 2)) return bt * betacf(a, b, x) / a;
   return 1 - bt * betacf(b, a, 1 - x) / b;
}

double pamlh_quantile_beta(double prob, double p, double q)
{
   /* bisection in log space near 0 (M7/M8 classes can be ~1e-30), then Newton polish */
   double lo = 0, hi = 1, x = 0.5;
   int i;
   for (i = 0; i < 400; i++) {
      double f;
      x = (lo > 0 && hi / lo > 4) ? sqrt(lo * hi) : 0.5 * (lo + hi);
      if (lo == 0 && hi < 1e-300) break;
      if (lo == 0 && i > 60) x = hi * 1e-3;
      f = pamlh_betai(p, q, x) - prob;
      if (f > 0) hi = x; else lo = x;
      if (hi - lo <= 1e-16 * hi) break;
   }
   return 0.5 * (lo + hi);
}
