/* The objective-function seam of the reference, on the engine:
 *
 *    double (*com.plfun)(double x[], int np);      codeml.c:125 / baseml.c:70 — returns MINUS lnL, minimised by ming2()
 *
 * A driver written against that pointer (the reference's own main() assigns lfun / lfundG / lfunAdG to it, codeml.c:2338) needs
 * nothing but a function of this type.  Below, the function is backed by libpamlh / libpaml_amd: SetParameters and the whole
 * likelihood evaluation happen behind it, on the GPU.
 *
 *    cc -I../../../include plfun_seam.c -L../../lib -lpamlh -lpaml_amd -lm -Wl,-rpath,'$ORIGIN/../../lib' -o plfun_seam
 *    ./plfun_seam codeml some/codeml.ctl
 */
#include <stdio.h>
#include <stdlib.h>

#include "pamlh.h"

static pamlh *g_analysis;                                   /* what `com` is to the reference: the data set and the model */

static double plfun_on_gpu(double x[], int np) { return pamlh_plfun(g_analysis, x, np); }

int main(int argc, char **argv)
{
   double (*plfun)(double x[], int np) = plfun_on_gpu;       /* <- the assignment a maintainer would make to com.plfun */
   double x[4096], f;
   char err[512];
   int np, i;
   if (argc < 3) { fprintf(stderr, "usage: %s <codeml|baseml> <ctl>\n", argv[0]); return 2; }
   if (pamlh_load(&g_analysis, argv[2], argv[1], err, sizeof(err))) { fprintf(stderr, "error: %s\n", err); return 1; }
   pamlh_dims(g_analysis, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, &np, NULL);
   pamlh_default_x(g_analysis, x, 4096);
   f = plfun(x, np);                                         /* one call of the objective function, as ming2 would make it */
   if (f >= 1e300) { fprintf(stderr, "error: %s\n", pamlh_error(g_analysis)); pamlh_free(g_analysis); return 1; }
   printf("-lnL = %.6f at the initial values (np = %d):", f, np);
   for (i = 0; i < np && i < 8; i++) printf(" %.4f", x[i]);
   printf("%s\n", np > 8 ? " ..." : "");
   pamlh_free(g_analysis);
   return 0;
}
