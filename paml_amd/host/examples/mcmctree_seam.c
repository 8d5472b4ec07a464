/* mcmctree_seam.c — the exact-likelihood consumer of mcmctree (usedata = 1) on the MI355X engine.
 *
 * In the reference, every MCMC proposal that moves node ages or rates ends in lnpD_locus (mcmctree.c:1130-1166): branch lengths
 * of the locus' gene tree from the ages and rates, then  lnL = -com.plfun(NULL, -1).  This program plays that loop for a chain of
 * (ages, rate) states read from stdin — "t_root t_.. ... mu" per line, the columns of the reference's mcmc.txt — and prints the
 * lnL of each: what mcmctree would have printed in its lnL column.
 *
 *    mcmctree_seam <baseml.ctl>  < states.txt
 * build: gcc -O2 -I include paml_amd/host/examples/mcmctree_seam.c -L paml_amd/lib -lpamlh -lpaml_amd -lm
 */
#include <stdio.h>
#include <stdlib.h>
#include "pamlh.h"

int main(int argc, char **argv)
{
   pamlh *p = NULL;
   char err[512];
   int n, ns, npatt, nn, root, i, np, ntime, first = 1;
   double x[4096], *age, mu, lnL;
   if (argc < 2) { fprintf(stderr, "usage: %s <baseml ctl with clock = 1>\n", argv[0]); return 2; }
   if (pamlh_load(&p, argv[1], "baseml", err, sizeof(err))) { fprintf(stderr, "error: %s\n", err); return 1; }
   pamlh_dims(p, &n, &ns, &npatt, &nn, &root, NULL, NULL, NULL, &np, &ntime);
   np = pamlh_default_x(p, x, 4096);                 /* substitution parameters from the control file; the ages are overridden */
   if (pamlh_set_x(p, x, np)) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
   age = (double *)calloc(nn, sizeof(double));
   for (;;) {
      for (i = ns; i < nn; i++)
         if (scanf("%lf", &age[i]) != 1) goto done;  /* internal nodes in the reference's numbering: root first */
      if (scanf("%lf", &mu) != 1) break;
      if (pamlh_lnpd_locus(p, age, mu, NULL, first, &lnL)) { fprintf(stderr, "error: %s\n", pamlh_error(p)); return 1; }
      first = 0;
      printf("%.3f\n", lnL);
   }
done:
   free(age);
   pamlh_free(p);
   return 0;
}
