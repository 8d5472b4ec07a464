/* pamlh_model.c — options -> data -> parameters -> engine inputs, the way the reference's model layer does it.
 * Written fresh against the behaviour of: GetOptions (codeml.c:1694, baseml.c:954), InitializeBaseAA (treesub.c:1548),
 * InitializeCodon / CountCodons / AddCodonFreqSeqGene (codeml.c:3640-3873), GetDaa (codeml.c:3967), eigenQcodon
 * (codeml.c:3229-3321), eigenQaa (3400-3484), eigenQREVbase / eigenTN93 (treesub.c:2488, 2210), SetParameters and
 * SetParametersNSsites (codeml.c:2757, 2459-2660; baseml.c:1306), DiscreteNSsites (codeml.c:2846), readx (treesub.c:4035). */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pamlh_internal.h"

/* genetic codes (codons in T, C, A, G order; '*' = stop) in the order of codeml's icode (GeneticCode[][] in tools.c:23-84): the NCBI
 * translation tables 1 (universal), 2 (vertebrate mt), 3 (yeast mt), 4 (mold mt), 5 (invertebrate mt), 6 (ciliate nuclear), 9 (echinoderm
 * mt), 10 (euplotid nuclear), 12 (alternative yeast nuclear), 13 (ascidian mt), 15 (blepharisma nuclear) — 60 to 63 sense codons.
 * icode 11 is the reference's "regularised" code: 64 sense codons, sixteen amino acids (ARNDCQEGHILKMFPST) with four codons each. */
#define N_GENETIC_CODES 12
static const char *const GENETIC_CODES[N_GENETIC_CODES] = {
   "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG", "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG",
   "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG", "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
   "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG", "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
   "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG", "FFLLSSSSYY**CCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
   "FFLLSSSSYY**CC*WLLLSPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG", "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG",
   "FFLLSSSSYY*QCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG", "RRRRNNNNDDDDCCCCQQQQEEEEGGGGHHHHIIIILLLLKKKKMMMMFFFFPPPPSSSSTTTT"};

const char *pamlh_genetic_code(int icode) { return icode >= 0 && icode < N_GENETIC_CODES ? GENETIC_CODES[icode] : NULL; }
static int nuc_nkappa(const pamlh *p);
/* days from 1970-01-01 to y-m-d in the proleptic Gregorian calendar (era arithmetic: 400-year cycles of 146 097 days, years starting in March) */
static long days_from_civil(int y, int m, int d)
{
   long era, yoe, doy, doe;
   y -= m <= 2;
   era = (y >= 0 ? y : y - 399) / 400;
   yoe = y - era * 400;
   doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
   doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
   return era * 146097 + doe - 719468;
}
static double dist2(const double *a, const double *b, int n)
{
   double s = 0;
   int i;
   for (i = 0; i < n; i++) s += (a[i] - b[i]) * (a[i] - b[i]);
   return sqrt(s);
}

/* base / amino-acid frequencies with ambiguity characters resolved by iteration (InitializeBaseAA treesub.c:1548-1700) */
static void add_freq(const pamlh *p, int js, const double *pi0, double *pi, int h0, int h1)
{
   int h, k, n = p->n;
   for (h = h0; h < h1; h++) {
      const int code = p->z[(size_t)js * p->npatt + h], nc = p->n_chara[code];
      const unsigned char *map = p->chara_map + (size_t)code * n;
      if (nc == 1) pi[map[0]] += p->w[h];
      else {
         double t = 0;
         for (k = 0; k < nc; k++) t += pi0[map[k]];
         for (k = 0; k < nc; k++) pi[map[k]] += p->w[h] * pi0[map[k]] / t;
      }
   }
}

/* frequencies of the patterns [h0, h1): mean over species of the per-species frequencies, then (with ambiguities) resolved
 * over all species together */
static void freqs_base_aa_range(pamlh *p, int h0, int h1, double *out)
{
   int n = p->n, js, k, it;
   double pi0[64], pi[64], piG[64] = {0}, t;
   for (js = 0; js < p->ns; js++) {
      for (k = 0; k < n; k++) pi0[k] = 1.0 / n;
      for (it = 0; it < 20; it++) {
         for (k = 0; k < n; k++) pi[k] = 0;
         add_freq(p, js, pi0, pi, h0, h1);
         for (k = 0, t = 0; k < n; k++) t += pi[k];
         for (k = 0; k < n; k++) pi[k] = t < 1e-10 ? 1.0 / n : pi[k] / t;
         if (p->cleandata || dist2(pi, pi0, n) < 1e-8) break;
         memcpy(pi0, pi, n * sizeof(double));
      }
      for (k = 0; k < n; k++) piG[k] += pi[k] / p->ns;
   }
   if (p->cleandata) { memcpy(out, piG, n * sizeof(double)); return; }
   memcpy(pi0, piG, n * sizeof(double));
   for (it = 0; it < 20; it++) {
      for (k = 0; k < n; k++) pi[k] = 0;
      for (js = 0; js < p->ns; js++) add_freq(p, js, pi0, pi, h0, h1);
      for (k = 0, t = 0; k < n; k++) t += pi[k];
      for (k = 0; k < n; k++) pi[k] /= t;
      if (dist2(pi, pi0, n) < 1e-8) break;
      memcpy(pi0, pi, n * sizeof(double));
   }
   memcpy(out, pi, n * sizeof(double));
}

static void freqs_base_aa(pamlh *p)
{
   const int n = p->n;
   int g, k, js, it;
   if (p->ngene <= 1) { freqs_base_aa_range(p, 0, p->npatt, p->pi_data); memcpy(p->piG[0], p->pi_data, n * sizeof(double)); return; }
   /* several genes (InitializeBaseAA treesub.c:1582-1668): com.piG[] within each gene; com.pi[] = their length-weighted mean
    * for clean data, else ambiguities resolved over the whole alignment starting from the plain mean of the genes */
   for (g = 0; g < p->ngene; g++) freqs_base_aa_range(p, p->posG[g], p->posG[g + 1], p->piG[g]);
   memset(p->pi_data, 0, sizeof(p->pi_data));
   if (p->cleandata) {
      for (g = 0; g < p->ngene; g++) for (k = 0; k < n; k++) p->pi_data[k] += p->piG[g][k] * p->lgene[g] / (double)p->ls;
   }
   else {
      double pi0[64] = {0}, pi[64], t;
      for (g = 0; g < p->ngene; g++) for (k = 0; k < n; k++) pi0[k] += p->piG[g][k] / p->ngene;
      for (it = 0; it < 20; it++) {
         for (k = 0; k < n; k++) pi[k] = 0;
         for (js = 0; js < p->ns; js++) add_freq(p, js, pi0, pi, 0, p->npatt);
         for (k = 0, t = 0; k < n; k++) t += pi[k];
         for (k = 0; k < n; k++) pi[k] /= t;
         if (dist2(pi, pi0, n) < 1e-8) break;
         memcpy(pi0, pi, n * sizeof(double));
      }
      memcpy(p->pi_data, pi, n * sizeof(double));
   }
}

static int base_set(char c, int *set)
{
   static const char BASEs[] = "TCAGUYRMKSWHBVD-N?";
   static const char *Eq[] = {"T", "C", "A", "G", "T", "TC", "AG", "CA", "TG", "CG", "TA", "TCA", "TCG", "CAG", "TAG", "TCAG", "TCAG", "TCAG"};
   const char *q = strchr(BASEs, c);
   int i, m = (int)strlen(Eq[q - BASEs]);
   for (i = 0; i < m; i++) set[i] = (int)(strchr(BASEs, Eq[q - BASEs][i]) - BASEs);
   return m;
}

/* codon data: fcodon (64), fb3x4, fb4 — CountCodons (resolved codons only) then the ambiguity iteration */
static void freqs_codon_range(pamlh *p, int h0, int h1, double *pi_out)
{
   int js, h, k, i0, i1, i2, it, np = p->npatt;
   double fc[64] = {0}, fb[12] = {0}, f4[4] = {0}, fc0[64], fb0[12], f40[4], t;
   int from64[64], nsense = 0;
   for (k = 0; k < 64; k++) from64[k] = p->code[k] == '*' ? -1 : nsense++;
   for (js = 0; js < p->ns; js++)
      for (h = h0; h < h1; h++) {
         const char *c = p->raw + ((size_t)js * np + h) * 3;
         int s[3][4], m[3];
         for (k = 0; k < 3; k++) m[k] = base_set(c[k], s[k]);
         if (m[0] * m[1] * m[2] > 1) continue;
         fc[s[0][0] * 16 + s[1][0] * 4 + s[2][0]] += p->w[h];
      }
   /* Fcodon_3x4: position-specific base frequencies from codon counts; fb4 = their average over positions */
   for (k = 0; k < 64; k++) { fb[k / 16] += fc[k]; fb[4 + (k / 4) % 4] += fc[k]; fb[8 + k % 4] += fc[k]; }
   for (k = 0; k < 3; k++) { for (t = 0, i0 = 0; i0 < 4; i0++) t += fb[k * 4 + i0]; for (i0 = 0; i0 < 4; i0++) fb[k * 4 + i0] /= t; }
   for (k = 0; k < 4; k++) f4[k] = (fb[k] + fb[4 + k] + fb[8 + k]) / 3;
   if (!p->cleandata) {
      for (t = 0, k = 0; k < 64; k++) t += fc[k];
      for (k = 0; k < 64; k++) fc0[k] = fc[k] / t;
      memcpy(fb0, fb, sizeof(fb)); memcpy(f40, f4, sizeof(f4));
      for (it = 0; it < 20; it++) {
         double d1, d2, d3;
         memset(fc, 0, sizeof(fc)); memset(fb, 0, sizeof(fb)); memset(f4, 0, sizeof(f4));
         for (js = 0; js < p->ns; js++)
            for (h = h0; h < h1; h++) {
               const char *c = p->raw + ((size_t)js * np + h) * 3;
               int s[3][4], m[3], ft[64] = {0}, nk = 0;
               double t1;
               for (k = 0; k < 3; k++) m[k] = base_set(c[k], s[k]);
               for (k = 0; k < 3; k++) {
                  for (i0 = 0, t = t1 = 0; i0 < m[k]; i0++) { t += fb0[k * 4 + s[k][i0]]; t1 += f40[s[k][i0]]; }
                  for (i0 = 0; i0 < m[k]; i0++) {
                     fb[k * 4 + s[k][i0]] += p->w[h] * fb0[k * 4 + s[k][i0]] / t;
                     f4[s[k][i0]] += p->w[h] * f40[s[k][i0]] / t1;
                  }
               }
               for (i0 = 0, t = 0; i0 < m[0]; i0++)
                  for (i1 = 0; i1 < m[1]; i1++)
                     for (i2 = 0; i2 < m[2]; i2++) {
                        const int ic = s[0][i0] * 16 + s[1][i1] * 4 + s[2][i2];
                        if (from64[ic] < 0) continue;
                        ft[ic] = 1; nk++; t += fc0[ic];
                     }
               for (k = 0; k < 64; k++)
                  if (ft[k]) fc[k] += (t > 0 ? p->w[h] * fc0[k] / t : p->w[h] / nk);
            }
         for (t = 0, k = 0; k < 64; k++) t += fc[k];
         for (k = 0; k < 64; k++) fc[k] /= t;
         for (k = 0; k < 3; k++) { for (t = 0, i0 = 0; i0 < 4; i0++) t += fb[k * 4 + i0]; for (i0 = 0; i0 < 4; i0++) fb[k * 4 + i0] /= t; }
         for (t = 0, k = 0; k < 4; k++) t += f4[k];
         for (k = 0; k < 4; k++) f4[k] /= t;
         d1 = dist2(fc, fc0, 64); d2 = dist2(fb, fb0, 12); d3 = dist2(f4, f40, 4);
         if (d1 < 1e-8 && d2 < 1e-8 && d3 < 1e-8) break;
         memcpy(fc0, fc, sizeof(fc)); memcpy(fb0, fb, sizeof(fb)); memcpy(f40, f4, sizeof(f4));
      }
   }
   memcpy(p->fcodon, fc, sizeof(fc)); memcpy(p->fb3x4, fb, sizeof(fb)); memcpy(p->fb4, f4, sizeof(f4));
   /* com.pi by CodonFreq (codeml.c:3852-3873) */
   {
      double s = 0;
      int j = 0;
      for (k = 0; k < 64; k++) {
         double v;
         if (from64[k] < 0) continue;
         if (p->codonfreq == 0) v = 1;
         else if (p->codonfreq == 1) v = f4[k / 16] * f4[(k / 4) % 4] * f4[k % 4];
         else if (p->codonfreq == 2) v = fb[k / 16] * fb[4 + (k / 4) % 4] * fb[8 + k % 4];
         else v = fc[k];
         pi_out[j++] = v;
         s += v;
      }
      for (j = 0; j < p->n; j++) pi_out[j] /= s;
   }
}

/* com.pi over all codons and com.piG within every gene (InitializeCodon codeml.c:3772-3900) */
static void freqs_codon(pamlh *p)
{
   int g;
   for (g = 0; g < p->ngene && p->ngene > 1; g++) {
      freqs_codon_range(p, p->posG[g], p->posG[g + 1], p->piG[g]);
      memcpy(p->fb3x4G[g], p->fb3x4, sizeof(p->fb3x4)); memcpy(p->fb4G[g], p->fb4, sizeof(p->fb4));
   }
   freqs_codon_range(p, 0, p->npatt, p->pi_data);      /* last: leaves the whole-data tables in fb3x4 / fb4 / fcodon */
   if (p->ngene <= 1) memcpy(p->piG[0], p->pi_data, p->n * sizeof(double));
}

static int read_aa_ratefile(pamlh *p)
{
   FILE *f = fopen(p->aaratefile, "r");
   int i, j;
   if (!f) return pamlh_fail(p, "cannot open aaRatefile %s", p->aaratefile);
   memset(p->aaS, 0, sizeof(p->aaS));
   for (i = 0; i < 20; i++)
      for (j = 0; j < i; j++) {
         double v;
         if (fscanf(f, "%lf", &v) != 1) { fclose(f); return pamlh_fail(p, "aaRatefile: too few rates"); }
         p->aaS[i * 20 + j] = p->aaS[j * 20 + i] = v;
      }
   for (i = 0; i < 20; i++)
      if (fscanf(f, "%lf", &p->aapi_file[i]) != 1) { fclose(f); return pamlh_fail(p, "aaRatefile: too few frequencies"); }
   fclose(f);
   return 0;
}

static void resolve(const pamlh *p, const char *name, char *out, size_t cap)
{
   if (name[0] == '/') snprintf(out, cap, "%s", name);
   else snprintf(out, cap, "%s/%s", p->dir, name);
}

static int read_omega_aa(pamlh *p);
static int read_aa_dist(pamlh *p);

static int aa_of_codon(const pamlh *p, int c64) { static const char AAS[] = "ARNDCQEGHILKMFPSTWYV"; return (int)(strchr(AAS, p->code[c64]) - AAS); }

/* codon frequencies from amino-acid frequencies, synonymous codons sharing equally (AA2Codonf codeml.c:3922) */
static void aa_to_codon_freqs(const pamlh *p, const double *faa, double *fc)
{
   int nsyn[20] = {0}, c, m = 0;
   for (c = 0; c < 64; c++) if (p->code[c] != '*') nsyn[aa_of_codon(p, c)]++;
   for (c = 0; c < 64; c++) if (p->code[c] != '*') { const int a = aa_of_codon(p, c); fc[m++] = faa[a] / nsyn[a]; }
}

/* model 5 (FromCodon0): from here on the data are codon data — every amino acid is the ambiguity code of its codons
 * (SetMapAmbiguity(.., 1) treesub.c:1274-1283, codeml.c:544-556), the frequencies those of aa_to_codon_freqs.  The reference
 * gives '?' / '-' an empty set there (such sites must be removed with cleandata = 1); here they stand for every codon. */
static void aa_as_codon_sets(pamlh *p)
{
   int from61[64], n = 0, c, a, k;
   size_t h;
   for (c = 0; c < 64; c++) if (p->code[c] != '*') from61[n++] = c;
   for (h = 0; h < (size_t)p->ns * p->npatt; h++) p->z[h] = (unsigned char)(n + (p->z[h] < 20 ? p->z[h] : 20));
   free(p->n_chara); free(p->chara_map);
   p->n_codes = n + 21;
   p->n_chara = (int *)calloc(p->n_codes, sizeof(int));
   p->chara_map = (unsigned char *)calloc((size_t)p->n_codes * n, 1);
   for (c = 0; c < n; c++) { p->n_chara[c] = 1; p->chara_map[(size_t)c * n] = (unsigned char)c; }
   for (a = 0; a < 20; a++)
      for (c = 0; c < n; c++) if (aa_of_codon(p, from61[c]) == a) p->chara_map[(size_t)(n + a) * n + p->n_chara[n + a]++] = (unsigned char)c;
   for (k = 0; k < n; k++) p->chara_map[(size_t)(n + 20) * n + k] = (unsigned char)k;
   p->n_chara[n + 20] = n;
   p->seqtype = 1; p->n = n; p->cleandata = 0; p->mg = 0;
   memcpy(p->pi_data, p->fb61, n * sizeof(double));
   memcpy(p->piG[0], p->fb61, n * sizeof(double));
}

int pamlh_load(pamlh **out, const char *ctl_path, const char *program, char *err, int errcap)
{
   return pamlh_load_tree(out, ctl_path, program, 0, err, errcap);
}

/* ... with the tree_index-th (0-based) tree of the tree file: the reference walks through all of them in turn (runmode = 0,
 * the loop over trees in Forestry codeml.c:635 / baseml.c:451); here every tree is an analysis of its own */
int pamlh_load_tree(pamlh **out, const char *ctl_path, const char *program, int tree_index, char *err, int errcap)
{
   return pamlh_load_with(out, ctl_path, program, tree_index, NULL, err, errcap);
}

/* ... and with control-file options replaced or added: `overrides` holds "key = value" lines (separated by newlines or ';').  A control
 * file that lists several site models ("NSsites = 0 1 2 7 8": the reference runs them one after the other, the insmodel loop codeml.c:657-906) is run here one
 * model per analysis: overrides = "NSsites = 2". */
int pamlh_load_with(pamlh **out, const char *ctl_path, const char *program, int tree_index, const char *overrides, char *err, int errcap)
{
   pamlh *p = (pamlh *)calloc(1, sizeof(pamlh));
   const char *v, *slash;
   int rc = 0;
   *out = NULL;
   p->itree = tree_index;
   p->is_codeml = strcmp(program, "baseml") != 0;
   slash = strrchr(ctl_path, '/');
   if (slash) { size_t n = (size_t)(slash - ctl_path); memcpy(p->dir, ctl_path, n); p->dir[n] = 0; }
   else strcpy(p->dir, ".");
   if ((rc = pamlh_read_ctl(p, ctl_path))) goto bad;
   if (overrides && (rc = pamlh_ctl_override(p, overrides))) goto bad;
   if (!(v = pamlh_opt(p, "seqfile"))) { rc = pamlh_fail(p, "no seqfile in the control file"); goto bad; }
   resolve(p, v, p->seqfile, sizeof(p->seqfile));
   if (!(v = pamlh_opt(p, "treefile"))) { rc = pamlh_fail(p, "no treefile in the control file"); goto bad; }
   resolve(p, v, p->treefile, sizeof(p->treefile));
   p->seqtype = p->is_codeml ? (int)pamlh_optd(p, "seqtype", 1) : 0;
   if (p->seqtype == 3) { p->translate = 1; p->seqtype = 2; p->icode = (int)pamlh_optd(p, "icode", 0); }      /* codons translated on reading, then an amino-acid analysis (ReadSeq treesub.c:886-892) */
   p->codonfreq = (int)pamlh_optd(p, "CodonFreq", 0);
   p->model = (int)pamlh_optd(p, "model", 0);
   p->nssites = (int)pamlh_optd(p, "NSsites", 0);
   if (p->nssites == 22) { p->m2a_rel = 1; p->nssites = 2; }      /* M2a_rel (NSM2aRel): M2a without the w2 > 1 constraint, the null of clade model C */
   p->icode = (int)pamlh_optd(p, "icode", 0);
   p->fix_kappa = (int)pamlh_optd(p, "fix_kappa", 0);
   p->kappa0 = pamlh_optd(p, "kappa", 2);
   p->fix_omega = (int)pamlh_optd(p, "fix_omega", 0);
   p->omega0 = pamlh_optd(p, "omega", 0.4);
   p->fix_alpha = (int)pamlh_optd(p, "fix_alpha", 1);
   p->alpha0 = pamlh_optd(p, "alpha", 0);
   p->ncatG = (int)pamlh_optd(p, "ncatG", 4);
   p->cleandata_opt = (int)pamlh_optd(p, "cleandata", 0);
   p->fix_blength = (int)pamlh_optd(p, "fix_blength", 0);
   /* options that would change the analysis and are not covered: refused, never ignored */
   {
      const int runmode = (int)pamlh_optd(p, "runmode", 0), ndata = (int)pamlh_optd(p, "ndata", 1);
      if (runmode != 0) { rc = pamlh_fail(p, "runmode = %d is not supported (0: the trees of the tree file; tree search and pairwise comparisons are outside this library)", runmode); goto bad; }
      if (ndata > 1) { rc = pamlh_fail(p, "ndata = %d is not supported (one data set per sequence file)", ndata); goto bad; }
      if ((int)pamlh_optd(p, "hkyREV", 0)) { rc = pamlh_fail(p, "hkyREV = 1 is not supported"); goto bad; }
      if ((int)pamlh_optd(p, "nparK", 0)) { rc = pamlh_fail(p, "nparK = %d is not supported", (int)pamlh_optd(p, "nparK", 0)); goto bad; }
      if ((int)pamlh_optd(p, "bootstrap", 0)) { rc = pamlh_fail(p, "bootstrap resampling of the alignment is not supported"); goto bad; }
   }
   p->fix_rho = (int)pamlh_optd(p, "fix_rho", 1);
   p->rho0 = pamlh_optd(p, "rho", 0);
   if (!p->fix_rho && p->rho0 == 0) p->rho0 = 0.001;      /* "init rho reset" (baseml.c:1087) */
   if (!p->fix_rho || p->rho0 != 0) {
      /* auto-discrete-gamma (lfunAdG): neighbouring sites' rate classes form a Markov chain with correlation rho */
      if (p->fix_alpha && p->alpha0 <= 0) { rc = pamlh_fail(p, "fix rho to 0 if alpha = 0"); goto bad; }
      if (p->nssites) { rc = pamlh_fail(p, "rho does not go with NSsites models"); goto bad; }
   }
   p->nhomo = p->is_codeml ? 0 : (int)pamlh_optd(p, "nhomo", 0);
   /* nhomo (baseml.c:1748-1760): 0 frequencies from the data, 1 as parameters, 2 a kappa per branch, 3 a frequency set for every
    * tip branch, one for the internal branches and one at the root, 4 a set for every node; 3 and 4 with a kappa per branch
    * (fix_kappa = 0), one kappa for all (fix_kappa = 1) or one per branch label (fix_kappa = 2); 5 the frequency sets by the '#'
    * labels of the tree file (the root takes a set of its own when it carries the next unused label). */
   if (p->nhomo < 0 || p->nhomo > 5) { rc = pamlh_fail(p, "nhomo = %d is not supported (0 ... 5)", p->nhomo); goto bad; }
   p->clock = (int)pamlh_optd(p, "clock", 0);
   if (p->clock < 0 || p->clock > 2) { rc = pamlh_fail(p, "clock = %d is not supported (0: no clock, 1: global clock, 2: local clocks by '#' rate labels in the tree)", p->clock); goto bad; }
   if ((v = pamlh_opt(p, "TipDate")) && atoi(v) != 0) {      /* "TipDate = 1 100": flag and time unit (GetOptions baseml.c / codeml.c) */
      double unit = -1;
      int flag = 0;
      sscanf(v, "%d %lf", &flag, &unit);
      p->tipdate = flag != 0; p->tip_timeunit = unit;
      if (!p->clock) p->tipdate = 0;      /* dates only matter to the clock models */
   }
   p->mgene = (int)pamlh_optd(p, "Mgene", 0);
   /* Mgene = 1 (separate analyses, MultipleGenes baseml.c:392 / codeml.c:570): the data set itself is not evaluated; every gene
    * is taken out as an analysis of its own with pamlh_gene_subset */
   if (p->mgene < 0 || p->mgene > 4) { rc = pamlh_fail(p, "Mgene = %d?", p->mgene); goto bad; }
   p->malpha = (int)pamlh_optd(p, "Malpha", 0) != 0;      /* a gamma shape per gene (checked against the data below) */
   if (p->seqtype == 1) {
      if (p->icode < 0 || p->icode >= N_GENETIC_CODES) { rc = pamlh_fail(p, "genetic code icode = %d is not supported (0 ... 11)", p->icode); goto bad; }
      /* model 0: site models; model 2, NSsites 0: branch model; model 2 / 3 with NSsites 2 / 3: branch-site A / B, clade C / D */
      if (p->model == 1 && p->nssites == 0) { p->free_ratio = 1; p->model = 2; p->fix_omega = 0; }      /* an omega for every branch: the branch model with one label per branch (below) */
      if (p->model != 0 && !(p->model == 2 && p->nssites == 0) && !((p->model == 2 || p->model == 3) && (p->nssites == 2 || p->nssites == 3))) {
         rc = pamlh_fail(p, "codon model = %d with NSsites = %d is not supported", p->model, p->nssites); goto bad;
      }
      if (p->m2a_rel && p->model) { rc = pamlh_fail(p, "NSsites = 22 (M2a_rel) is a site model (model = 0)"); goto bad; }
      if (p->model == 2 && p->nssites == 3 && p->fix_omega) { rc = pamlh_fail(p, "fix_omega with branch-site model B is not supported"); goto bad; }
      if (p->model && p->nssites && (p->alpha0 > 0 || !p->fix_alpha)) { rc = pamlh_fail(p, "dN/dS ratios among branches are not supported with gamma rates"); goto bad; }
      if (p->model == 3) p->ncatG = 3;                /* "ncatG = 3 reset" (codeml.c:1607) */
      strcpy(p->code, GENETIC_CODES[p->icode]);
      if (p->nssites == 4) p->ncatG = 5;      /* M4 (freqs): omega = 0, 1/3, 2/3, 1, 3 with free proportions */
      if (p->nssites != 0 && p->nssites != 1 && p->nssites != 2 && p->nssites != 3 && p->nssites != 4 && p->nssites != 5 && p->nssites != 6 && p->nssites != 7 && p->nssites != 8 && !(p->nssites >= 9 && p->nssites <= 13)) { rc = pamlh_fail(p, "NSsites = %d is not supported", p->nssites); goto bad; }
      if (p->model == 0 && p->nssites == 3 && (p->fix_omega || p->ncatG < 2 || p->ncatG > 16)) { rc = pamlh_fail(p, "NSsites = 3 needs fix_omega = 0 and 2 <= ncatG <= 16"); goto bad; }
      if (p->codonfreq < 0 || p->codonfreq > 7) { rc = pamlh_fail(p, "CodonFreq = %d is not supported", p->codonfreq); goto bad; }
      p->codonf_model = p->codonfreq;
      p->est_freq = (int)pamlh_optd(p, "estFreq", 0) != 0;
      /* F1x4MG / F3x4MG (4, 5): the frequencies of F1x4 / F3x4, Muse-Gaut style rates (GetMutationMultiplier codeml.c:3060);
       * FMutSel0 / FMutSel (6, 7; Yang & Nielsen 2008): mutation bias pi_T, pi_C, pi_A (always parameters) and amino-acid / codon
       * fitnesses — estimated (estFreq = 1) or implied by the observed frequencies; the codon table is the observed one */
      if (p->codonfreq >= 6) { p->mg = 1; p->mutsel = p->codonfreq - 5; p->codonfreq = 3; }
      else if (p->codonfreq >= 4) { p->mg = 1; p->codonfreq -= 3; }
      /* frequency parameters in x (GetOptions codeml.c:1574-1588) */
      if (p->mutsel == 1) p->npi = 3 + (p->est_freq ? 19 : 0);
      else if (p->mutsel == 2) p->npi = 3 + (p->est_freq ? -1 : 0);      /* + ncode - 1 once the code is known (below) */
      else if (p->est_freq && p->codonfreq) p->npi = p->codonfreq == 1 ? 3 : p->codonfreq == 2 ? 9 : -1;
      for (p->n = 0, rc = 0; rc < 64; rc++) p->n += p->code[rc] != '*';
      rc = 0;
      if (p->npi == 2) p->npi = 3 + p->n - 1;      /* FMutSel with estFreq: the codon fitnesses */
      else if (p->npi == -1) p->npi = p->n - 1;    /* Fcodon with estFreq */
      p->aadist = (int)pamlh_optd(p, "aaDist", 0);
      if (p->aadist < -6 || p->aadist > 7) { rc = pamlh_fail(p, "aaDist = %d is not supported (1..6 / -1..-6: distance files, 7: AAClasses)", p->aadist); goto bad; }
      if (p->aadist == 7) {
         if (p->nssites || (p->model != 0 && p->model != 2) || p->mg) { rc = pamlh_fail(p, "aaDist = 7 goes with NSsites = 0, model 0 or 2 and CodonFreq <= 3"); goto bad; }
         if ((rc = read_omega_aa(p))) goto bad;
      }
      else if (p->aadist) {
         /* omega_ij = b exp(-a d_ij) (aaDist > 0) or b (1 - a d_ij) (< 0), d = the chosen amino-acid distance over its maximum */
         if (p->nssites || p->model || p->mg) { rc = pamlh_fail(p, "aaDist & NSsites / model don't work together"); goto bad; }
         if (p->fix_omega) { rc = pamlh_fail(p, "can't fix_omega for aaDist models"); goto bad; }
         if ((rc = read_aa_dist(p))) goto bad;
      }
   }
   else if (p->seqtype == 2) {
      p->n = 20; p->aa_model = p->model;
      if ((p->aa_model < 0 || p->aa_model > 3) && p->aa_model != 5 && p->aa_model != 6 && p->aa_model != 8 && p->aa_model != 9) { rc = pamlh_fail(p, "amino-acid model %d is not supported", p->aa_model); goto bad; }
      if (p->aa_model >= 8) {
         /* REVaa_0 (8): an exchangeability for every pair of amino acids one nucleotide change apart under the genetic code, the others 0;
          * REVaa (9): all 190; the pair V-I is the unit (ijAAref codeml.c:1091, SetAA1STEP 4044, eigenQaa 3423-3436) */
         int i, j, c1, c2, step[400] = {0};
         if (p->icode < 0 || p->icode >= N_GENETIC_CODES) { rc = pamlh_fail(p, "genetic code icode = %d is not supported (0 ... 11)", p->icode); goto bad; }
         strcpy(p->code, GENETIC_CODES[p->icode]);
         for (c1 = 0; c1 < 64; c1++)
            for (c2 = 0; c2 < c1; c2++) {
               const int nd = (c1 / 16 != c2 / 16) + ((c1 / 4) % 4 != (c2 / 4) % 4) + (c1 % 4 != c2 % 4);
               if (nd == 1 && p->code[c1] != '*' && p->code[c2] != '*') { const int a = aa_of_codon(p, c1), b = aa_of_codon(p, c2); step[a * 20 + b] = step[b * 20 + a] = 1; }
            }
         memset(p->aa1step, 0, sizeof(p->aa1step));
         for (p->n_aarate = 0, i = 1; i < 20; i++)
            for (j = 0; j < i; j++)
               if ((p->aa_model == 9 || step[i * 20 + j]) && i * 20 + j != 19 * 20 + 9) { p->aa1step[i * 20 + j] = 1; p->n_aarate++; }
         if ((v = pamlh_opt(p, "aaRatefile")) && *v) {      /* initial values */
            resolve(p, v, p->aaratefile, sizeof(p->aaratefile));
            if ((rc = read_aa_ratefile(p))) goto bad;
         }
         else { for (i = 0; i < 400; i++) p->aaS[i] = 1; }
      }
      if (p->aa_model >= 5) {
         /* codon-based amino-acid models (Yang, Nielsen & Hasegawa 1998): 6 (FromCodon) a 20-state chain whose rates are the
          * codon chain's aggregated over synonymous codons, 5 (FromCodon0) the codon chain itself with every amino acid read as
          * the set of its codons (codeml.c:1513-1531, 498-503, 544-556) */
         if (p->icode < 0 || p->icode >= N_GENETIC_CODES) { rc = pamlh_fail(p, "genetic code icode = %d is not supported (0 ... 11)", p->icode); goto bad; }
         if (p->nssites) { rc = pamlh_fail(p, "use NSsites = 0 for amino acids"); goto bad; }
         if ((int)pamlh_optd(p, "aaDist", 0)) { rc = pamlh_fail(p, "aaDist with the codon-based amino-acid models is not supported"); goto bad; }
         if (p->aa_model == 6 && p->fix_omega) { rc = pamlh_fail(p, "fix_omega = 1?  omega is not estimable!"); goto bad; }
         if (p->clock) { rc = pamlh_fail(p, "model and clock do not work together"); goto bad; }
         strcpy(p->code, GENETIC_CODES[p->icode]);
         p->codonfreq = 0;       /* "CodonFreq=0 reset": the codon frequencies come from the amino-acid frequencies */
         p->model = 0;
      }
      if (p->aa_model == 2 || p->aa_model == 3) {
         if (!(v = pamlh_opt(p, "aaRatefile")) || !*v) { rc = pamlh_fail(p, "empirical aa model without aaRatefile"); goto bad; }
         resolve(p, v, p->aaratefile, sizeof(p->aaratefile));
         if ((rc = read_aa_ratefile(p))) goto bad;
      }
   }
   else if (p->seqtype == 0) {
      p->n = 4;
      if (p->model > UNREST) { rc = pamlh_fail(p, "baseml model %d is not supported", p->model); goto bad; }
   }
   else { rc = pamlh_fail(p, "seqtype %d is not supported", p->seqtype); goto bad; }
   if ((rc = pamlh_read_seqs(p))) goto bad;
   if ((rc = pamlh_read_tree(p))) goto bad;
   if (p->nhomo == 1 && (p->ngene > 1 || p->model < F81 || p->model > REV)) { rc = pamlh_fail(p, "nhomo = 1 needs one gene and a model with base frequencies (F81 ... REV)"); goto bad; }
   if (p->nhomo >= 2) {
      if (p->ngene > 1) { rc = pamlh_fail(p, "nhomo for several genes?"); goto bad; }
      if (p->nhomo == 2 && p->model != K80 && p->model != F84 && p->model != HKY85) { rc = pamlh_fail(p, "nhomo = 2 works with K80, F84 or HKY85"); goto bad; }
      if (p->nhomo > 2 && (p->model < F84 || p->model > REV)) { rc = pamlh_fail(p, "nhomo = %d needs a model with base frequencies and rate parameters (F84 ... REV)", p->nhomo); goto bad; }
      if (p->nhomo > 2 && (p->fix_kappa < 0 || p->fix_kappa > 2)) { rc = pamlh_fail(p, "nhomo: fix_kappa = %d?", p->fix_kappa); goto bad; }
      if (p->nnode > PAMLH_MAXEIG) { rc = pamlh_fail(p, "nhomo >= 2: more than %d nodes", PAMLH_MAXEIG); goto bad; }
      if (!p->fix_rho || p->rho0 != 0) { rc = pamlh_fail(p, "nhomo with rho is not supported"); goto bad; }
   }
   if (p->malpha && (p->ngene <= 1 || p->fix_alpha || !(p->alpha0 > 0) || p->mgene == 1)) {
      if (p->mgene == 1 && p->ngene > 1) p->malpha = 0;      /* separate analyses: every gene has its own alpha anyway */
      else { rc = pamlh_fail(p, "Malpha needs several genes (option G) and a free alpha > 0"); goto bad; }
   }
   if (p->malpha && (!p->fix_rho || p->rho0 != 0)) { rc = pamlh_fail(p, "Malpha or rho"); goto bad; }
   if (p->ngene <= 1) { if (p->mgene) { rc = pamlh_fail(p, "Mgene = %d but the sequence file has one gene (no option G)", p->mgene); goto bad; } }
   else {
      /* what the several-gene set-up covers (the reference's own exclusions: baseml.c:261-265, codeml.c:1534-1544) */
      if (p->fix_blength >= 2) { rc = pamlh_fail(p, "fix_blength = 2 or 3 does not work for partitioned data"); goto bad; }
      if (!p->fix_rho || p->rho0 != 0) { rc = pamlh_fail(p, "rho with several genes is not supported"); goto bad; }
      if (p->seqtype == 1 && (p->model || p->nssites)) { rc = pamlh_fail(p, "several genes: only the one-ratio codon model (model 0, NSsites 0)"); goto bad; }
      if (p->mgene == 1) goto genes_ok;
      if (p->mgene >= 3 && (p->fix_kappa || (p->seqtype == 1 && p->fix_omega))) { rc = pamlh_fail(p, "Mgene = %d needs free kappa (and omega)", p->mgene); goto bad; }
      if (p->seqtype == 2 && p->mgene >= 3) { rc = pamlh_fail(p, "Mgene = %d has no meaning for the amino-acid models here", p->mgene); goto bad; }
      if (p->seqtype == 2 && p->mgene == 2 && p->aa_model != 3 && p->aa_model != 1) { rc = pamlh_fail(p, "Mgene = 2 needs frequencies from the data (amino-acid model 1 or 3)"); goto bad; }
      if (p->seqtype == 1 && p->mgene == 2 && p->codonfreq == 0) { rc = pamlh_fail(p, "Mgene = 2 with equal codon frequencies"); goto bad; }
      if (p->seqtype == 0 && p->mgene >= 2 && p->model == UNREST) { rc = pamlh_fail(p, "Mgene >= 2 does not work with UNREST"); goto bad; }
      if (p->seqtype == 0 && ((p->mgene >= 2 && p->model == JC69) || (p->mgene >= 3 && p->model == F81) || ((p->mgene == 2 || p->mgene == 4) && p->model == K80))) {
         rc = pamlh_fail(p, "this Mgene option has no meaning for the model"); goto bad;
      }
   }
genes_ok:
   if (p->clock == 2) { p->rate_label = (int *)malloc(p->nnode * sizeof(int)); memcpy(p->rate_label, p->label, p->nnode * sizeof(int)); }      /* local clocks: '#' = rate class */
   if (p->seqtype == 0 && p->nhomo >= 3) {      /* nhomo 5 / fix_kappa 2: '#' = set (ReadTreeN treesub.c:8878-8885: branch types = largest label + 1, the root aside) */
      int v;
      p->nh_label = (int *)malloc(p->nnode * sizeof(int)); memcpy(p->nh_label, p->label, p->nnode * sizeof(int));
      for (p->nh_nbtype = 0, v = 0; v < p->nnode; v++) if (v != p->root && p->nh_label[v] + 1 > p->nh_nbtype) p->nh_nbtype = p->nh_label[v] + 1;
      if (p->nhomo == 5 && (p->nh_label[p->root] < 0 || p->nh_label[p->root] > p->nh_nbtype)) { rc = pamlh_fail(p, "nhomo = 5: label for root strange?"); goto bad; }
   }
   if (!(p->seqtype == 1 && p->model >= 2)) memset(p->label, 0, p->nnode * sizeof(int));      /* '#' labels only matter to branch models */
   if (p->seqtype == 0 && p->nhomo >= 2) { int v; for (v = 0; v < p->nnode; v++) p->label[v] = v; }      /* every branch has its own P(t) family */
   if (p->free_ratio) { int b; for (b = 0; b < p->nbranch; b++) p->label[p->branch_node[b]] = b; }      /* free-ratio model: nodes[tree.branches[i][1]].label = i (codeml.c:2171-2175) */
   if (p->seqtype == 1 && p->model >= 2) {
      int i;
      for (p->n_omega = 1, i = 0; i < p->nnode; i++) if (p->label[i] + 1 > p->n_omega) p->n_omega = p->label[i] + 1;
      if (p->nssites && p->model == 2 && p->n_omega != 2) { rc = pamlh_fail(p, "branch-site models need two branch types (label the foreground branches #1), the tree has %d", p->n_omega); goto bad; }
      if (p->nssites && p->model == 3 && (p->n_omega < 2 || p->n_omega > 16)) { rc = pamlh_fail(p, "clade models need 2..16 branch types, the tree has %d", p->n_omega); goto bad; }
   }
   if (p->seqtype == 1) freqs_codon(p);
   else freqs_base_aa(p);
   if (p->seqtype == 1 && p->npi) {
      int c, m = 0;
      if (p->ngene > 1) { rc = pamlh_fail(p, "codon models (estFreq) not implemented for ngene > 1"); goto bad; }
      if (p->aadist) { rc = pamlh_fail(p, "estFreq / FMutSel with aaDist is not supported"); goto bad; }
      memset(p->pi_aa, 0, sizeof(p->pi_aa));
      for (c = 0; c < 64; c++) if (p->code[c] != '*') p->pi_aa[aa_of_codon(p, c)] += p->pi_data[m++];
   }
   if (p->seqtype == 2 && p->aa_model >= 5) {
      if (p->ngene > 1) { rc = pamlh_fail(p, "the codon-based amino-acid models take one gene"); goto bad; }
      aa_to_codon_freqs(p, p->pi_data, p->fb61);
      if (p->aa_model == 5) aa_as_codon_sets(p);
   }
   if (p->seqtype == 0 && p->model == T92) {      /* one GC-content parameter: T = A, C = G (InitializeBaseAA treesub.c:1684-1691) */
      int g;
      p->pi_data[0] = p->pi_data[2] = (p->pi_data[0] + p->pi_data[2]) / 2; p->pi_data[1] = p->pi_data[3] = (p->pi_data[1] + p->pi_data[3]) / 2;
      for (g = 0; g < p->ngene; g++) {
         p->piG[g][0] = p->piG[g][2] = (p->piG[g][0] + p->piG[g][2]) / 2; p->piG[g][1] = p->piG[g][3] = (p->piG[g][1] + p->piG[g][3]) / 2;
      }
   }
   /* parameter bookkeeping (GetInitials): ntime, np */
   p->ntime = p->fix_blength == 2 ? 0 : p->fix_blength == 3 ? 1 : p->nbranch;      /* 3: the tree file's lengths times one free factor */
   if (p->clock) {
      /* global clock (SetBranch treesub.c:3793-3809, GetInitialsTimes 3814): a rooted binary tree, the parameters are the ages
       * of the ns - 1 internal nodes (in node order; this is the reference's layout once LASTROUND is set), tips at age 0 */
      if (p->fix_blength >= 2) { rc = pamlh_fail(p, "clock with fix_blength = %d", p->fix_blength); goto bad; }
      if (p->sons_ptr[p->root + 1] - p->sons_ptr[p->root] != 2 || p->nnode != 2 * p->ns - 1) { rc = pamlh_fail(p, "clock = 1 needs a rooted binary tree"); goto bad; }
      if (p->seqtype == 1 && p->model) { rc = pamlh_fail(p, "model and clock do not work together"); goto bad; }
      p->ntime = p->ns - 1;
      if (p->clock == 2) {
         /* local clocks (GetInitialsTimes treesub.c:3866-3886, GetBranchRate 3678-3700): the '#' labels of the tree file are rate
          * classes — class 0 runs at rate 1, the rates of classes 1 .. nbtype-1 follow the ages in x; the length of a branch is its
          * time span times the rate of its class */
         int v;
         if (p->nhomo >= 2) { rc = pamlh_fail(p, "clock = 2 and nhomo are incompatible"); goto bad; }
         for (p->n_brate = 1, v = 0; v < p->nnode; v++) if (v != p->root && p->rate_label[v] + 1 > p->n_brate) p->n_brate = p->rate_label[v] + 1;
         if (p->n_brate <= 1) { rc = pamlh_fail(p, "use clock = 1 or add branch rate labels (#1 ...) in the tree"); goto bad; }
         for (v = 1; v < p->n_brate; v++) {
            int w, found = 0;
            for (w = 0; w < p->nnode; w++) if (w != p->root && p->rate_label[w] == v) found = 1;
            if (!found) { rc = pamlh_fail(p, "not all branch rate labels 0 ... %d are on the tree", p->n_brate - 1); goto bad; }
         }
         p->ntime += p->n_brate - 1;
      }
      if (p->tipdate) {
         /* TipDate (GetTipDate treesub.c:3552-3622, GetAgeLow 3750, GetBranchRate 3678): the last field of every sequence name is its
          * sampling date; tip age = (youngest date - date) / time unit; x = node ages, then the mutation rate per time unit (the rate
          * of branch class 0), then the other classes' rates; a node cannot be younger than the oldest tip below it */
         double young = 0, old = 0;
         int v, changed = 1;
         p->tip_age = (double *)calloc(p->nnode, sizeof(double));
         p->age_low = (double *)calloc(p->nnode, sizeof(double));
         for (v = 0; v < p->ns; v++) {
            const char *nm = p->names[v], *q = nm + strlen(nm);
            double d = 0;
            while (q > nm && (isdigit((unsigned char)q[-1]) || q[-1] == '.' || q[-1] == '-')) q--;
            if (strchr(q, '-')) {
               /* yyyy-mm-dd (or yyyy-mm: the 15th): days since 1970-01-01.  The reference takes mktime() / 86400 (treesub.c:3573-3582),
                * i.e. the same count shifted by the local time zone; differences of dates are what enters the model */
               int y = 0, mo = 0, dd = 0, nf = sscanf(q, "%d-%d-%d", &y, &mo, &dd);
               if (nf < 2 || mo < 1 || mo > 12) { rc = pamlh_fail(p, "TipDate: date format wrong in the name %s (yyyy-mm-dd)", nm); goto bad; }
               if (nf < 3 || dd < 1) dd = 15;
               d = (double)days_from_civil(y, mo, dd);
            }
            else if (!*q || sscanf(q, "%lf", &d) != 1) d = 0;
            if (d <= 0) { rc = pamlh_fail(p, "TipDate: no sampling date at the end of the name %s", nm); goto bad; }
            p->tip_age[v] = d;
            if (v == 0 || d > young) young = d;
            if (v == 0 || d < old) old = d;
         }
         if (young - old < 1e-100) { rc = pamlh_fail(p, "TipDate: all sequences are of the same age?"); goto bad; }
         if (p->tip_timeunit <= 0) p->tip_timeunit = (young - old) * 2.5;
         for (v = 0; v < p->ns; v++) { p->tip_age[v] = (young - p->tip_age[v]) / p->tip_timeunit; if (p->tip_age[v] < 1e-100) p->tip_age[v] = 0; p->age_low[v] = p->tip_age[v]; }
         while (changed)
            for (changed = 0, v = 0; v < p->nnode; v++)
               if (v != p->root && p->age_low[p->father[v]] < p->age_low[v]) { p->age_low[p->father[v]] = p->age_low[v]; changed = 1; }
         p->ntime += 1;
      }
   }
   {
      int nr = p->ngene - 1;       /* rgene */
      const int rep = p->mgene >= 3 ? p->ngene : 1;
      if (p->seqtype == 1) {
         nr += !p->fix_kappa;
         nr += p->npi;
         if (p->aadist == 7) nr += p->n_omega_type * (p->model == 2 ? p->n_omega : 1);      /* AAClasses: a set of class omegas (per branch label) */
         else if (p->aadist) nr += 2;                                                          /* a, b of omega(d) */
         else if (p->nssites == 0 && p->model == 2) nr += p->n_omega - (p->fix_omega != 0);      /* branch model: one omega per branch label, the last one fixed under fix_omega (codeml.c:2170-2183) */
         else if (p->model == 2 && p->nssites == 2) nr += 3 + !p->fix_omega;      /* branch-site A: p0 p1 w0 [w2] (codeml.c:2197-2221) */
         else if (p->model == 2 && p->nssites == 3) nr += 5;                      /* branch-site B: p0 p1 w0 w1 w2 */
         else if (p->model == 3) nr += 2 + (p->nssites == 3 ? 2 : 1) + p->n_omega - (p->fix_omega != 0);   /* clade C / D (codeml.c:2222-2233) */
         else if (p->nssites == 3) nr += 2 * p->ncatG - 1;                        /* M3: K-1 proportions, K omegas */
         else if (p->nssites == 4) nr += 4;                                       /* M4: 4 proportions */
         else if (p->nssites == 5) nr += 2;                                       /* M5: gamma(a, b) */
         else if (p->nssites == 6) nr += 4;                                       /* M6: p0, a1, b1, a2 */
         else if (p->nssites >= 9 && p->nssites <= 12) nr += 5;                   /* M9 - M12: five parameters (CDFdN_dS codeml.c:2925-2934) */
         else if (p->nssites == 13) nr += 6;                                      /* M13: p0, p1, mu2, s0, s1, s2 */
         else if (p->nssites == 0) nr += !p->fix_omega;
         else if (p->nssites == 1) nr += 2;
         else if (p->nssites == 2) nr += 3 + !p->fix_omega;      /* M2a: p0 p1 w0 [w2]; fix_omega fixes w2 (omega_fix, codeml.c:1596) */
         else if (p->nssites == 7) nr += 2;
         else if (p->nssites == 8) nr += 3 + !p->fix_omega;
      }
      else if (p->seqtype == 2 && p->aa_model == 6) nr += !p->fix_kappa;
      else if (p->seqtype == 2 && p->aa_model >= 8) nr += p->n_aarate;
      else if (p->seqtype == 0) {
         if (p->model == K80 || p->model == HKY85 || p->model == F84 || p->model == T92) nr += !p->fix_kappa;
         else if (p->model == TN93) nr += 2 * !p->fix_kappa;
         else if (p->model == REV) nr += 5;
         else if (p->model == UNREST) nr += 11;
         if (p->nhomo == 1) nr += p->model == T92 ? 1 : 3;
         if (p->nhomo >= 2) nr = pamlh_nh_nrate(p) + (p->nhomo > 2 ? pamlh_nh_npi(p) * (p->model == T92 ? 1 : 3) : 0);
      }
      if (rep > 1) nr += (rep - 1) * (p->seqtype == 1 ? 2 : nuc_nkappa(p));      /* Mgene 3, 4: a parameter set per gene */
      if (p->alpha0 > 0 || !p->fix_alpha) nr += p->malpha ? p->ngene : !p->fix_alpha;
      nr += !p->fix_rho;
      p->np = p->ntime + nr;
   }
   p->branch = (double *)calloc(p->nnode, sizeof(double));
   p->pi = (double *)calloc(64 * PAMLH_MAXGENE, sizeof(double));
   p->freqK = (double *)calloc(64, sizeof(double));
   p->rate = (double *)calloc(64 * PAMLH_MAXGENE, sizeof(double));
   p->eigen_of = (int *)calloc(64 * PAMLH_MAXEIG, sizeof(int));
   p->n_pi = 1;
   *out = p;
   return 0;
bad:
   if (err && errcap > 0) snprintf(err, errcap, "%s", p->err);
   pamlh_free(p);
   return rc ? rc : -1;
}

void pamlh_free(pamlh *p)
{
   if (p) free(p->pose);
   int i;
   if (!p) return;
   if (p->eng) paml_amd_destroy(p->eng);
   if (p->names) for (i = 0; i < p->ns; i++) free(p->names[i]);
   free(p->names); free(p->z); free(p->w); free(p->raw); free(p->n_chara); free(p->chara_map);
   free(p->rate_label); free(p->nh_label); free(p->tip_age); free(p->age_low);
   free(p->sons_ptr); free(p->sons); free(p->label); free(p->branch_node); free(p->father); free(p->tree_branch); free(p->scale);
   free(p->branch); free(p->pi); free(p->freqK); free(p->rate); free(p->eigen_of);
   for (i = 0; i < PAMLH_MAXEIG; i++) pamlh_eig_release(&p->eig[i]);
   free(p->gene_eigen_of);
   free(p);
}

const char *pamlh_error(const pamlh *p) { return p ? p->err : "null"; }

int pamlh_dims(const pamlh *p, int *n_states, int *n_tips, int *n_patt, int *n_nodes, int *root, int *n_codes,
               int *cleandata, int *ls, int *np, int *ntime)
{
   if (n_states) *n_states = p->n;
   if (n_tips) *n_tips = p->ns;
   if (n_patt) *n_patt = p->npatt;
   if (n_nodes) *n_nodes = p->nnode;
   if (root) *root = p->root;
   if (n_codes) *n_codes = p->n_codes;
   if (cleandata) *cleandata = p->cleandata;
   if (ls) *ls = p->ls;
   if (np) *np = p->np;
   if (ntime) *ntime = p->ntime;
   return 0;
}
const unsigned char *pamlh_tips(const pamlh *p) { return p->z; }
const double *pamlh_weights(const pamlh *p) { return p->w; }
const int *pamlh_n_chara(const pamlh *p) { return p->n_chara; }
const unsigned char *pamlh_chara_map(const pamlh *p) { return p->chara_map; }
const int *pamlh_sons_ptr(const pamlh *p) { return p->sons_ptr; }
const int *pamlh_sons(const pamlh *p) { return p->sons; }
const int *pamlh_labels(const pamlh *p) { return p->label; }
const unsigned char *pamlh_scale_nodes(const pamlh *p) { return p->scale; }
const int *pamlh_branch_order(const pamlh *p) { return p->branch_node; }
const double *pamlh_branch(const pamlh *p) { return p->branch; }
const double *pamlh_pi(const pamlh *p) { return p->pi; }
const double *pamlh_freqK(const pamlh *p) { return p->freqK; }
const double *pamlh_rate(const pamlh *p) { return p->rate; }
const int *pamlh_eigen_of(const pamlh *p) { return p->eigen_of; }

int pamlh_model(const pamlh *p, int *mode, int *K, int *n_eigen, int *n_labels)
{
   if (mode) *mode = p->mode;
   if (K) *K = p->K;
   if (n_eigen) *n_eigen = p->n_eigen;
   if (n_labels) *n_labels = p->n_labels;
   return 0;
}

int pamlh_eigen(const pamlh *p, int i, int *kind, int *nR, double *kappa, const double **U, const double **V,
                const double **Root, const double **Cijk)
{
   if (i < 0 || i >= p->n_eigen) return -1;
   if (U || V || Root) pamlh_eig_host((pamlh_eig *)&p->eig[i], p->n);      /* (formed on demand when the device does the decompositions) */
   if (kind) *kind = p->eig[i].kind;
   if (nR) *nR = p->eig[i].nR;
   if (kappa) *kappa = p->eig[i].kappa;
   if (U) *U = p->eig[i].U;
   if (V) *V = p->eig[i].V;
   if (Root) *Root = p->eig[i].Root;
   if (Cijk) *Cijk = p->eig[i].Cijk;
   return 0;
}

/* ---- codon frequencies as parameters (estFreq = 1) and FMutSel0 / FMutSel ----------------------------------------------------------
 * x after kappa (GetInitials codeml.c:2107-2134): pi_T / pi_G, pi_C / pi_G, pi_A / pi_G (F1x4 types and both FMutSel models; one triple
 * per codon position for the F3x4 types), then — FMutSel0 with estFreq — 19 amino-acid fitnesses, or — FMutSel / Fcodon with
 * estFreq — ncode - 1 codon fitnesses (logs, the last amino acid / codon at 0).  GetCodonFreqs (codeml.c:2690-2755) turns them into
 * com.pf3x4 and com.pi. */
static int codon_freq_initials(const pamlh *p, double *x)
{
   int k = 0, i, j, c, m;
   if (p->mutsel || p->codonfreq == 1) for (i = 0; i < 3; i++) x[k++] = p->fb4[i] / p->fb4[3];
   else if (p->codonfreq == 2) for (j = 0; j < 3; j++) for (i = 0; i < 3; i++) x[k++] = p->fb3x4[j * 4 + i] / p->fb3x4[j * 4 + 3];
   if (p->mutsel == 1 && p->npi > 3) {
      int nsyn[20] = {0};
      for (c = 0; c < 64; c++) if (p->code[c] != '*') nsyn[aa_of_codon(p, c)]++;
      for (i = 0; i < 19; i++) x[k++] = log((p->pi_aa[i] / nsyn[i] + .001) / (p->pi_aa[19] / nsyn[19] + .001));
   }
   else if ((p->mutsel == 2 && p->npi > 3) || (!p->mutsel && p->codonfreq == 3))
      for (m = 0; m < p->n - 1; m++) x[k++] = log((p->pi_data[m] + .001) / (p->pi_data[p->n - 1] + .001));
   return k;
}

static void codon_freqs_from_x(pamlh *p, const double *ppi)
{
   const int n = p->n;
   int from61[64], m = 0, c, i, j, b[3];
   double t;
   for (c = 0; c < 64; c++) if (p->code[c] != '*') from61[m++] = c;
   if (!p->mutsel && p->codonfreq == 3) {      /* Fcodon */
      for (t = 0, i = 0; i < n; i++) { p->pi[i] = i == n - 1 ? 1 : exp(ppi[i]); t += p->pi[i]; }
      for (i = 0; i < n; i++) p->pi[i] /= t;
      return;
   }
   for (j = 0; j < 3; j++) {
      for (t = 1, i = 0; i < 3; i++) { p->pf3x4[j * 4 + i] = ppi[i]; t += ppi[i]; }
      p->pf3x4[j * 4 + 3] = 1;
      for (i = 0; i < 4; i++) p->pf3x4[j * 4 + i] /= t;
      if (!p->mutsel && p->codonfreq == 2) ppi += 3;
   }
   if (p->mutsel == 2 && p->npi == 3) { memcpy(p->pi, p->pi_data, n * sizeof(double)); return; }      /* the observed codon table */
   if (p->mutsel == 1 && p->npi == 3) {      /* amino-acid frequencies as observed, synonymous codons by the mutation bias */
      double mutbias[20] = {0};
      for (i = 0; i < n; i++) { c = from61[i]; mutbias[aa_of_codon(p, c)] += p->pf3x4[c / 16] * p->pf3x4[(c / 4) % 4] * p->pf3x4[c % 4]; }
      for (i = 0; i < n; i++) { c = from61[i]; p->pi[i] = p->pf3x4[c / 16] * p->pf3x4[(c / 4) % 4] * p->pf3x4[c % 4] / mutbias[aa_of_codon(p, c)] * p->pi_aa[aa_of_codon(p, c)]; }
   }
   else {
      for (i = 0; i < n; i++) { c = from61[i]; b[0] = c / 16; b[1] = (c / 4) % 4; b[2] = c % 4; p->pi[i] = p->pf3x4[b[0]] * p->pf3x4[4 + b[1]] * p->pf3x4[8 + b[2]]; }
      if (p->mutsel == 2) for (i = 0; i < n - 1; i++) p->pi[i] *= exp(ppi[3 + i]);
      else if (p->mutsel == 1) for (i = 0; i < n; i++) { const int a = aa_of_codon(p, from61[i]); if (a < 19) p->pi[i] *= exp(ppi[3 + a]); }
   }
   for (t = 0, i = 0; i < n; i++) t += p->pi[i];
   for (i = 0; i < n; i++) p->pi[i] /= t;
}

int pamlh_default_x(const pamlh *p, double *x, int cap)
{
   int k = 0, i;
   if (cap < p->np) return -1;
   if (p->fix_blength == 3) x[k++] = 1;
   else if (p->clock) {         /* ages: 0.04 per level above the deepest tip */
      int node, changed = 1, *hgt = (int *)calloc(p->nnode, sizeof(int));
      while (changed)
         for (changed = 0, node = 0; node < p->nnode; node++)
            if (node != p->root && hgt[p->father[node]] < hgt[node] + 1) { hgt[p->father[node]] = hgt[node] + 1; changed = 1; }
      for (node = p->ns; node < p->nnode; node++) x[k++] = 0.04 * hgt[node];
      if (p->tipdate) {      /* above the oldest tip below: the root 1.5 times that age (at least 0.1 more), the others 60 % of the way up */
         int changed2 = 1, *done = (int *)calloc(p->nnode, sizeof(int));
         x[p->root - p->ns] = p->age_low[p->root] * 1.5 + 0.1; done[p->root] = 1;
         while (changed2)
            for (changed2 = 0, node = p->ns; node < p->nnode; node++)
               if (!done[node] && done[p->father[node]]) { x[node - p->ns] = p->age_low[node] + (x[p->father[node] - p->ns] - p->age_low[node]) * 0.6; done[node] = 1; changed2 = 1; }
         free(done);
         x[k++] = 0.1;      /* mutation rate per time unit */
      }
      free(hgt);
      for (node = 1; node < (p->clock == 2 ? p->n_brate : 1); node++) x[k++] = p->tipdate ? 0.1 : 1;      /* branch rates */
   }
   else
   for (i = 0; i < p->ntime; i++) { double b = p->tree_branch[p->branch_node[i]]; x[k++] = b >= 0 ? b : 0.1; }
   for (i = 1; i < p->ngene; i++) x[k++] = 1;      /* rgene */
   if (p->ngene > 1 && p->mgene >= 3) {            /* a parameter set per gene */
      int g, j;
      for (g = 0; g < p->ngene; g++) {
         if (p->seqtype == 1) { x[k++] = p->kappa0; x[k++] = p->omega0; }
         else for (j = 0; j < nuc_nkappa(p); j++) x[k++] = p->model == REV ? 1 : p->kappa0;
      }
      if (!p->fix_alpha) { int ga; for (ga = 0; ga < (p->malpha ? p->ngene : 1); ga++) x[k++] = p->alpha0 > 0 ? p->alpha0 : 0.5; }
      return k;
   }
   if (p->seqtype == 1) {
      if (!p->fix_kappa) x[k++] = p->kappa0;
      if (p->npi) k += codon_freq_initials(p, x + k);
      if (p->aadist == 7) { for (i = 0; i < p->n_omega_type * (p->model == 2 ? p->n_omega : 1); i++) x[k++] = 0.15 + 0.02 * (i % 4); }
      else if (p->aadist) { x[k++] = 0.15; x[k++] = 0.25; }
      else if (p->nssites == 0 && p->model == 2) { for (i = 0; i < p->n_omega - (p->fix_omega != 0); i++) x[k++] = p->fix_omega ? 0.4 : p->omega0; }
      else if (p->model == 2 && p->nssites) {      /* branch-site A / B: p0 p1 w0 [w1] [w2] */
         x[k++] = 0.6; x[k++] = 0.2; x[k++] = 0.25;
         if (p->nssites == 3) x[k++] = 0.8;
         if (p->nssites == 3 || !p->fix_omega) x[k++] = p->omega0 > 1 ? p->omega0 : 2.5;
      }
      else if (p->model == 3) {                     /* clade C / D: p0 p1 w0 [w1] then one omega per branch type */
         x[k++] = 0.5; x[k++] = 0.3; x[k++] = 0.25;
         if (p->nssites == 3) x[k++] = 0.75;
         for (i = 0; i < p->n_omega - (p->fix_omega != 0); i++) x[k++] = p->omega0 * (1 + 0.25 * i);
      }
      else if (p->nssites == 4) { for (i = 0; i < 4; i++) x[k++] = 0.2; }
      else if (p->nssites == 5) { x[k++] = 0.5; x[k++] = 1.0; }
      else if (p->nssites == 6) { x[k++] = 0.6; x[k++] = 0.5; x[k++] = 1.5; x[k++] = 2.0; }
      else if (p->nssites == 9 || p->nssites == 10) { x[k++] = 0.8; x[k++] = 0.5; x[k++] = 1.5; x[k++] = 1.0; x[k++] = 1.0; }
      else if (p->nssites == 11) { x[k++] = 0.8; x[k++] = 0.5; x[k++] = 1.5; x[k++] = 2.0; x[k++] = 1.0; }
      else if (p->nssites == 12) { x[k++] = 0.3; x[k++] = 0.6; x[k++] = 2.0; x[k++] = 0.5; x[k++] = 1.0; }
      else if (p->nssites == 13) { x[k++] = 0.4; x[k++] = 0.4; x[k++] = 2.0; x[k++] = 0.3; x[k++] = 0.5; x[k++] = 1.0; }
      else if (p->nssites == 3) {                   /* M3: K-1 proportions, K omegas */
         for (i = 0; i < p->ncatG - 1; i++) x[k++] = 1.0 / p->ncatG;
         for (i = 0; i < p->ncatG; i++) x[k++] = 0.1 + 1.4 * i / (p->ncatG - 1);
      }
      else if (p->nssites == 0) { if (!p->fix_omega) x[k++] = p->omega0; }
      else if (p->nssites == 1) { x[k++] = 0.6; x[k++] = 0.1; }
      else if (p->nssites == 2) { x[k++] = 0.5; x[k++] = 0.3; x[k++] = 0.1; if (!p->fix_omega) x[k++] = 2.5; }
      else if (p->nssites == 7) { x[k++] = 0.5; x[k++] = 1.5; }
      else if (p->nssites == 8) { x[k++] = 0.9; x[k++] = 0.5; x[k++] = 1.5; if (!p->fix_omega) x[k++] = 2.5; }
   }
   else if (p->seqtype == 2) {
      if (p->aa_model == 6 && !p->fix_kappa) x[k++] = p->kappa0;
      if (p->aa_model >= 8) { int a, b; for (a = 1; a < 20; a++) for (b = 0; b < a; b++) if (p->aa1step[a * 20 + b]) { const double u = p->aaS[19 * 20 + 9] > 0 ? p->aaS[a * 20 + b] / p->aaS[19 * 20 + 9] : 1; x[k++] = u > 1e-3 ? u : 1e-3; } }
   }
   else if (p->seqtype == 0) {
      if ((p->model == K80 || p->model == HKY85 || p->model == F84 || p->model == T92) && !p->fix_kappa) x[k++] = p->kappa0;
      else if (p->model == TN93 && !p->fix_kappa) { x[k++] = p->kappa0; x[k++] = p->kappa0; }
      else if (p->model == REV) { for (i = 0; i < 5; i++) x[k++] = 1; }
      else if (p->model == UNREST) { for (i = 0; i < 11; i++) x[k++] = (i == 0 || i == 3 || i == 8) ? 0.9 : 0.5; }
      if (p->nhomo == 1) { if (p->model == T92) x[k++] = p->pi_data[1] + p->pi_data[3]; else for (i = 0; i < 3; i++) x[k++] = p->pi_data[i]; }
      if (p->nhomo >= 2) {
         int j;
         k = p->ntime;
         for (i = 0; i < pamlh_nh_nrate(p); i++) x[k++] = p->model == REV ? 1 : p->kappa0;
         for (j = 0; j < (p->nhomo > 2 ? pamlh_nh_npi(p) : 0); j++) { if (p->model == T92) x[k++] = p->pi_data[1] + p->pi_data[3]; else for (i = 0; i < 3; i++) x[k++] = p->pi_data[i]; }
      }
   }
   if (!p->fix_alpha) { int ga; for (ga = 0; ga < (p->malpha ? p->ngene : 1); ga++) x[k++] = p->alpha0 > 0 ? p->alpha0 : 0.5; }
   if (!p->fix_rho) x[k++] = p->rho0;
   return k;
}

int pamlh_read_inx(const pamlh *p, double *x, int cap)
{
   char path[1200];
   FILE *f;
   int k = 0;
   double v;
   snprintf(path, sizeof(path), "%s/%s", p->dir, p->is_codeml ? "in.codeml" : "in.baseml");
   if (!(f = fopen(path, "r"))) return 0;
   if (fscanf(f, "%lf", &v) == 1) {
      if (v != -1 && k < cap) x[k++] = v;    /* a leading -1 = "evaluate at exactly these values" (treesub.c:4057) */
      while (k < cap && fscanf(f, "%lf", &v) == 1) x[k++] = v;
   }
   fclose(f);
   return k;
}

/* Where the decomposition of Q = S diag(pi) happens.  20- and 61-state models: on the DEVICE — the host keeps Q, pi and the scale,
 * and the upload (pamlh_upload_eigen_sets) hands all waiting matrices to paml_amd_set_eigen_qrev_batch in one call; U, V, Root are
 * formed on the host only when something asks for them (pamlh_eig_host: the accessor of the oracle-based tests, tables).
 * 4 states (and PAMLH_HOST_EIGEN=1, the A/B switch): on the host at once, as before. */
static int host_eigen_runtime;      /* set when a device decomposition did not converge (PAML_AMD_ENOCONV): from then on the host decomposes */
static int host_eigen_forced(void)
{
   static int v = -1;
   if (v < 0) v = getenv("PAMLH_HOST_EIGEN") != NULL;
   return v || host_eigen_runtime;
}
int pamlh_force_host_eigen(void)      /* returns 0 when the host was decomposing already (nothing left to fall back to) */
{
   if (host_eigen_forced()) return 0;
   fprintf(stderr, "pamlh: a device eigen-decomposition did not converge: the rate matrices are decomposed on the host from here on\n");
   host_eigen_runtime = 1;
   return 1;
}

void pamlh_eig_release(pamlh_eig *e)
{
   free(e->U); free(e->V); free(e->Root); free(e->Cijk); free(e->Q); free(e->qpi); free(e->qv);
   e->U = e->V = e->Root = e->Cijk = e->Q = e->qpi = e->qv = NULL;
   e->qv_n = 0;
   e->lazy = 0;
}

void pamlh_eig_host(pamlh_eig *e, int n)
{
   int k;
   if (!e->lazy || !e->Q) return;
   if (!e->U) { e->U = (double *)malloc((size_t)n * n * 8); e->V = (double *)malloc((size_t)n * n * 8); e->Root = (double *)malloc(n * 8); }
   pamlh_eigen_qrev(e->Q, e->qpi, n, e->Root, e->U, e->V);
   for (k = 0; k < n; k++) e->Root[k] /= e->scale;
   e->lazy = 0;
}

static int codon_pattern(const pamlh *p, const int **row, const int **col);
static void set_eig_uvroot(pamlh *p, int i, const double *Q, const double *pi, double scale)
{
   const int n = p->n;
   pamlh_eig *e = &p->eig[i];
   e->kind = PAML_AMD_EIGEN_UVROOT;
   if (!e->Q) { e->Q = (double *)malloc((size_t)n * n * 8); e->qpi = (double *)malloc(n * 8); }
   memcpy(e->Q, Q, (size_t)n * n * 8);
   memcpy(e->qpi, pi, n * 8);
   e->scale = scale;
   e->lazy = 1;
   {
      const int *row, *col;
      const int nnz = codon_pattern(p, &row, &col);
      int k;
      if (nnz && !e->qv) e->qv = (double *)malloc(704 * sizeof(double));
      for (k = 0; k < nnz; k++) e->qv[k] = Q[row[k] * n + col[k]];
      e->qv_n = nnz;
   }
   if (n <= 5 || host_eigen_forced()) pamlh_eig_host(e, n);
}

/* The eigen systems of the model state go to the engine as sets base, base + 1, ...: reversible rate matrices still waiting for
 * their decomposition are collected — into `batch` when the caller gathers several model states for one device call
 * (pamlh_eig_batch_flush), else into a batch of their own — everything else is sent as it is. */
/* Per genetic code (a model's code never changes; several models of a process may differ in theirs): the pairs of sense codons one
 * nucleotide apart, i > j — the only off-diagonal elements codon_q_cls fills: 263 under the universal code — and, for the hand-over of a
 * rate matrix as its elements (paml_amd_set_eigen_qrev_batch_sparse), those positions and the diagonal in row-major order. */
typedef struct {
   int ready, npair, nnz;
   int i[640], j[640], c1[640], c2[640], row[704], col[704];
   int nbr_ptr[65], nbr[1280];      /* row by row, ascending: the columns a row has elements in (its neighbours one nucleotide away) */
} codon_pairs_t;
static const codon_pairs_t *codon_pairs(const pamlh *p)
{
   static codon_pairs_t tab[32];
   static char codes[32][65];
   int slot, k, i, j;
   codon_pairs_t *t = NULL;
#pragma omp critical(pamlh_codon_pairs)
   {
      for (slot = 0; slot < 32 && tab[slot].ready; slot++)
         if (!memcmp(codes[slot], p->code, 64)) break;
      if (slot == 32) slot = 31;      /* (more codes than there are: the last slot is made again) */
      t = &tab[slot];
      if (!t->ready || memcmp(codes[slot], p->code, 64)) {
         int from61[64], m = 0;
         for (k = 0; k < 64; k++) if (p->code[k] != '*') from61[m++] = k;
         t->npair = t->nnz = 0;
         for (i = 0; i < m; i++)
            for (j = 0; j <= i; j++) {
               const int c1 = from61[i], c2 = from61[j];
               const int nd = (c1 / 16 != c2 / 16) + ((c1 / 4) % 4 != (c2 / 4) % 4) + (c1 % 4 != c2 % 4);
               if (nd == 1) { t->i[t->npair] = i; t->j[t->npair] = j; t->c1[t->npair] = c1; t->c2[t->npair++] = c2; }
               if (nd <= 1) { t->row[t->nnz] = i; t->col[t->nnz++] = j; }
            }
         for (i = 0, k = 0; i < m; i++) {
            t->nbr_ptr[i] = k;
            for (j = 0; j < m; j++) {
               const int c1 = from61[i], c2 = from61[j];
               if ((c1 / 16 != c2 / 16) + ((c1 / 4) % 4 != (c2 / 4) % 4) + (c1 % 4 != c2 % 4) == 1) t->nbr[k++] = j;
            }
         }
         t->nbr_ptr[m] = k;
         memcpy(codes[slot], p->code, 64);
         t->ready = 1;
      }
   }
   return t;
}

/* The elements a codon rate matrix can have, at and below its diagonal (above).  PAMLH_DENSE_Q=1: the matrices travel whole, as they
 * did before round 6; PAMLH_CHECK_SPARSE=1: every matrix is looked through for anything outside. */
static int codon_pattern(const pamlh *p, const int **row, const int **col)
{
   static int dense = -1;
   const codon_pairs_t *cp;
   int m = 0, k;
   if (dense < 0) dense = getenv("PAMLH_DENSE_Q") != NULL;
   if (dense || p->seqtype != 1 || p->n < 21) return 0;
   for (k = 0; k < 64; k++) m += p->code[k] != '*';
   if (m != p->n) return 0;
   cp = codon_pairs(p);
   *row = cp->row; *col = cp->col;
   return cp->nnz;
}

static void eig_batch_add(const pamlh *p, pamlh_eig_batch *b, int id, const pamlh_eig *e, int n)
{
   if (!b->cnt) b->nnz = codon_pattern(p, &b->row, &b->col);
   {
      const size_t per = b->nnz ? (size_t)b->nnz : (size_t)n * n;
      if (b->cnt == b->cap) {
         b->cap = b->cap ? 2 * b->cap : 64;
         b->ids = (int *)realloc(b->ids, b->cap * sizeof(int));
         b->Q = (double *)realloc(b->Q, (size_t)b->cap * per * 8);
         b->pi = (double *)realloc(b->pi, (size_t)b->cap * n * 8);
         b->scale = (double *)realloc(b->scale, b->cap * 8);
      }
      b->n = n;
      b->ids[b->cnt] = id;
      if (b->nnz) {
         double *v = b->Q + (size_t)b->cnt * per;
         int k;
         if (e->qv_n == b->nnz) memcpy(v, e->qv, (size_t)b->nnz * 8);
         else for (k = 0; k < b->nnz; k++) v[k] = e->Q[b->row[k] * n + b->col[k]];
         static int check = -1;
         if (check < 0) check = getenv("PAMLH_CHECK_SPARSE") != NULL;
         if (check) {
            int i, j, at = 0;
            for (i = 0; i < n; i++)
               for (j = 0; j <= i; j++) {
                  if (at < b->nnz && b->row[at] == i && b->col[at] == j) { at++; continue; }
                  if (e->Q[i * n + j] != 0) { fprintf(stderr, "pamlh: a codon matrix has an element at (%d, %d) the pattern does not\n", i, j); abort(); }
               }
         }
      }
      else memcpy(b->Q + (size_t)b->cnt * per, e->Q, per * 8);
      memcpy(b->pi + (size_t)b->cnt * n, e->qpi, n * 8);
      b->scale[b->cnt++] = e->scale;
   }
}

/* Where a batch of rate matrices is decomposed: on the device from PAMLH_DEVICE_EIGEN_MIN matrices on.  Round 3 left batches of fewer
 * than 16 (the 1-3 eigen systems per trial point of M0 / M1a / M2a) to the host's cores — one cold Jacobi run on the device costs
 * ~1.2 ms however many matrices ride in it, a core's Householder + QL 0.56 ms.  With the warm start a search now switches on
 * (pamlh_optimize; 3-5 sweeps instead of 9-10) the device is as fast for the small batches too — time to the MLEs, MI355X,
 * threshold 16 / 4 / 1: HIV M1a 0.05 / 0.04 / 0.04 s, M7 0.13 / 0.13 / 0.11, M8 0.23 / 0.21 / 0.21, everything else equal
 * (profiles/r04_time_to_mle.txt) — and ONE path decomposes every matrix of a run: the base point of a gradient and its perturbed
 * points no longer come from two algorithms whose U, V, Root differ in the last bits.  Default 1. */
static int device_eigen_min(void)
{
   static int v = -1;
   if (v < 0) { const char *e = getenv("PAMLH_DEVICE_EIGEN_MIN"); v = e ? atoi(e) : 1; }
   return v;
}

int pamlh_eig_batch_flush(pamlh *p, paml_amd_engine *eng, pamlh_eig_batch *b)
{
   int rc = 0;
   if (b->cnt && b->nnz && b->cnt < device_eigen_min()) {      /* (the host's solver wants whole matrices) */
      const int n = b->n, m = b->cnt;
      double *full = (double *)calloc((size_t)m * n * n, 8);
      int i, k;
      for (i = 0; i < m; i++)
         for (k = 0; k < b->nnz; k++) {
            const double v = b->Q[(size_t)i * b->nnz + k];
            full[((size_t)i * n + b->row[k]) * n + b->col[k]] = v;
            full[((size_t)i * n + b->col[k]) * n + b->row[k]] = v;      /* (only the lower triangle is read) */
         }
      free(b->Q); b->Q = full; b->nnz = 0;
   }
   if (b->cnt && b->cnt < device_eigen_min()) {
      const int n = b->n, m = b->cnt;
      double *uvr = (double *)malloc((size_t)m * (2 * n * n + n) * sizeof(double));
      int i, k;
#pragma omp parallel for schedule(dynamic) num_threads(m < 16 ? m : 16) if (m > 1)
      for (i = 0; i < m; i++) {
         double *U = uvr + (size_t)i * (2 * n * n + n), *V = U + (size_t)n * n, *R = V + (size_t)n * n;
         int kk;
         pamlh_eigen_qrev(b->Q + (size_t)i * n * n, b->pi + (size_t)i * n, n, R, U, V);
         for (kk = 0; kk < n; kk++) R[kk] /= b->scale[i];
      }
      for (i = 0; i < m && !rc; i++) {
         const double *U = uvr + (size_t)i * (2 * n * n + n);
         rc = paml_amd_set_eigen_uvroot(eng, b->ids[i], U, U + (size_t)n * n, U + (size_t)2 * n * n);
      }
      (void)k;
      free(uvr);
      if (rc) pamlh_fail(p, "%s", paml_amd_last_error(eng));
   }
   else if (b->cnt && (rc = b->nnz ? paml_amd_set_eigen_qrev_batch_sparse(eng, b->cnt, b->ids, b->nnz, b->row, b->col, b->Q, b->pi, b->scale)
                                    : paml_amd_set_eigen_qrev_batch(eng, b->cnt, b->ids, b->Q, b->pi, b->scale))) pamlh_fail(p, "%s", paml_amd_last_error(eng));
   free(b->ids); free(b->Q); free(b->pi); free(b->scale);
   memset(b, 0, sizeof(*b));
   return rc;
}

int pamlh_upload_eigen_sets(pamlh *p, paml_amd_engine *eng, int base, pamlh_eig_batch *batch)
{
   pamlh_eig_batch own;
   int i, rc = 0;
   memset(&own, 0, sizeof(own));
   for (i = 0; i < p->n_eigen && !rc; i++) {
      pamlh_eig *e = &p->eig[i];
      if (e->kind == PAML_AMD_EIGEN_UVROOT && e->lazy && e->Q) eig_batch_add(p, batch ? batch : &own, base + i, e, p->n);
      else if (e->kind == PAML_AMD_EIGEN_UVROOT) rc = paml_amd_set_eigen_uvroot(eng, base + i, e->U, e->V, e->Root);
      else if (e->kind == PAML_AMD_EIGEN_CIJK) rc = paml_amd_set_eigen_cijk(eng, base + i, e->nR, e->Cijk, e->Root);
      else if (e->kind == PAML_AMD_EIGEN_K80) rc = paml_amd_set_eigen_k80(eng, base + i, e->kappa);
      else if (e->kind == PAML_AMD_EIGEN_QMAT) rc = paml_amd_set_eigen_qmat(eng, base + i, e->U);
      else rc = paml_amd_set_eigen_jc69like(eng, base + i);
   }
   if (rc) { pamlh_fail(p, "%s", paml_amd_last_error(eng)); free(own.ids); free(own.Q); free(own.pi); free(own.scale); return rc; }
   return batch ? 0 : pamlh_eig_batch_flush(p, eng, &own);
}

/* aaDist = 7 (AAClasses): OmegaAA.dat beside the control file — "nclass", then for classes 1 .. nclass-1 a line "k: XY XY ...",
 * every other pair that can change in one step being class 0 (GetOmegaAA codeml.c:4079-4150; pairs that cannot change in one
 * step under the genetic code are ignored, a pair listed twice is an error). */
static int read_omega_aa(pamlh *p)
{
   char path[1200], line[4096];
   FILE *f;
   int i, j, k, c1, c2, ncls = 0, cls;
   memset(p->omega_class, -1, sizeof(p->omega_class));
   for (c1 = 0; c1 < 64; c1++)            /* AA1STEP: amino acids joined by a single nucleotide change between sense codons */
      for (c2 = 0; c2 < 64; c2++) {
         int nd = (c1 / 16 != c2 / 16) + ((c1 / 4) % 4 != (c2 / 4) % 4) + (c1 % 4 != c2 % 4);
         if (nd == 1 && p->code[c1] != '*' && p->code[c2] != '*' && p->code[c1] != p->code[c2])
            p->omega_class[p->code[c1] - 'A'][p->code[c2] - 'A'] = 0;
      }
   snprintf(path, sizeof(path), "%s/OmegaAA.dat", p->dir);
   if (!(f = fopen(path, "r"))) return pamlh_fail(p, "aaDist = 7 needs OmegaAA.dat beside the control file (%s)", path);
   if (fscanf(f, "%d", &ncls) != 1 || ncls < 1 || ncls > 64) { fclose(f); return pamlh_fail(p, "OmegaAA.dat: bad number of classes"); }
   for (cls = 1; cls < ncls; cls++) {
      if (fscanf(f, "%d", &k) != 1 || k != cls || fgetc(f) != ':' || !fgets(line, sizeof(line), f)) { fclose(f); return pamlh_fail(p, "OmegaAA.dat: expected \"%d: pairs\"", cls); }
      for (i = 0; line[i] && line[i] != '\n'; i++) {
         if (!isalpha((unsigned char)line[i])) continue;
         j = i + 1;
         if (!isalpha((unsigned char)line[j])) { fclose(f); return pamlh_fail(p, "OmegaAA.dat: amino acids come in pairs"); }
         c1 = toupper((unsigned char)line[i]) - 'A'; c2 = toupper((unsigned char)line[j]) - 'A';
         i = j;
         if (p->omega_class[c1][c2] == -1) continue;      /* cannot change in one step: ignored, as the reference does */
         if (p->omega_class[c1][c2] > 0) { fclose(f); return pamlh_fail(p, "OmegaAA.dat: pair %c%c listed twice", c1 + 'A', c2 + 'A'); }
         p->omega_class[c1][c2] = p->omega_class[c2][c1] = (signed char)cls;
      }
   }
   fclose(f);
   p->n_omega_type = ncls;
   return 0;
}

/* aaDist 1 .. 6 (and their negatives): grantham.dat, miyata.dat, g1974c / p / v / a.dat beside the control file — the lower triangle
 * of a 20 x 20 table in the order ARNDCQEGHILKMFPSTWYV, divided by its largest entry (GetDaa codeml.c:3967-3993) */
static int read_aa_dist(pamlh *p)
{
   static const char *const files[] = {"", "grantham.dat", "miyata.dat", "g1974c.dat", "g1974p.dat", "g1974v.dat", "g1974a.dat"};
   static const char AAS[] = "ARNDCQEGHILKMFPSTWYV";
   char path[1200];
   double d[400], dmax = 0;
   int i, j;
   FILE *f;
   snprintf(path, sizeof(path), "%s/%s", p->dir, files[abs(p->aadist)]);
   if (!(f = fopen(path, "r"))) return pamlh_fail(p, "aaDist = %d needs %s beside the control file", p->aadist, files[abs(p->aadist)]);
   for (i = 0; i < 20; i++)
      for (j = 0, d[i * 20 + i] = 0; j < i; j++) {
         if (fscanf(f, "%lf", &d[i * 20 + j]) != 1) { fclose(f); return pamlh_fail(p, "%s: too few distances", path); }
         d[j * 20 + i] = d[i * 20 + j];
         if (d[i * 20 + j] > dmax) dmax = d[i * 20 + j];
      }
   fclose(f);
   memset(p->aa_dist, 0, sizeof(p->aa_dist));
   for (i = 0; i < 20; i++) for (j = 0; j < 20; j++) p->aa_dist[AAS[i] - 'A'][AAS[j] - 'A'] = d[i * 20 + j] / dmax;
   return 0;
}

/* codon Q for (kappa, omega) and its mean rate (eigenQcodon codeml.c:3274-3315) */
static double codon_q_cls(const pamlh *p, const double *pi, double kappa, double omega, const double *wcls, double *Q);
static double codon_q_pi(const pamlh *p, const double *pi, double kappa, double omega, double *Q) { return codon_q_cls(p, pi, kappa, omega, NULL, Q); }
static double codon_q(const pamlh *p, double kappa, double omega, double *Q) { return codon_q_cls(p, p->pi, kappa, omega, NULL, Q); }
/* wcls != NULL: aaDist = 7, omega of a nonsynonymous change = wcls[class of its amino-acid pair]; aaDist 1..6 / -1..-6, wcls = (a, b) of
 * omega = b exp(-a d) / b (1 - a d) (GetOmega codeml.c:3020) */
static double codon_q_cls(const pamlh *p, const double *pi, double kappa, double omega, const double *wcls, double *Q)
{
   /* the pairs of sense codons one nucleotide apart (i > j): 263 of the 1 830 pairs under the universal code — from the genetic code's
    * table, not found again for every matrix of every trial point */
   const codon_pairs_t *cp = codon_pairs(p);
   const int npair = cp->npair, *pi_ = cp->i, *pj_ = cp->j, *pc1 = cp->c1, *pc2 = cp->c2;
   int i, j, k, n = p->n, x;
   double mr = 0;
   memset(Q, 0, (size_t)n * n * sizeof(double));
   for (x = 0; x < npair; x++) {
         const int c1 = pc1[x], c2 = pc2[x];
         const int f[3] = {c1 / 16, (c1 / 4) % 4, c1 % 4}, t[3] = {c2 / 16, (c2 / 4) % 4, c2 % 4};
         int pos = 0;
         double q = 1;
         i = pi_[x]; j = pj_[x];
         for (k = 0; k < 3; k++) if (f[k] != t[k]) pos = k;
         if (f[pos] + t[pos] == 1 || f[pos] + t[pos] == 5) q = kappa;
         if (p->mg) {      /* divide by the frequencies of the two unchanged nucleotides: the rate depends on the target nucleotide only */
            const int b1 = (pos + 1) % 3, b2 = (pos + 2) % 3;
            if (p->npi) q /= p->pf3x4[b1 * 4 + t[b1]] * p->pf3x4[b2 * 4 + t[b2]];      /* the tables in effect (parameters) */
            else q /= (p->codonfreq == 2 ? p->fb3x4[b1 * 4 + t[b1]] * p->fb3x4[b2 * 4 + t[b2]] : p->fb4[t[b1]] * p->fb4[t[b2]]);
            if (p->mutsel) {
               /* fixation probability of a mutant of fitness F_j in a population of F_i, up to a constant (Yang & Nielsen 2008 eq. 2-4;
                * GetMutationMultiplier codeml.c:3074-3084): e^F = pi / (mutation-bias product) */
               const double small = 1e-6 < 1.0 / p->ls ? 1e-6 : 1.0 / p->ls;
               double e1 = (pi[i] > small ? pi[i] : small) / (p->pf3x4[f[0]] * p->pf3x4[f[1]] * p->pf3x4[f[2]]);
               double e2 = (pi[j] > small ? pi[j] : small) / (p->pf3x4[t[0]] * p->pf3x4[t[1]] * p->pf3x4[t[2]]);
               if (fabs(e2 - e1) > 1e-10) q *= (log(e2) - log(e1)) / (e2 - e1);
               else q /= e2;
            }
         }
         if (p->code[c1] != p->code[c2]) {
            const int a1 = p->code[c1] - 'A', a2 = p->code[c2] - 'A';
            if (!wcls) q *= omega;
            else if (p->aadist == 7) q *= wcls[(int)p->omega_class[a1][a2]];
            else { const double w = wcls[0] * p->aa_dist[a1][a2]; q *= (p->aadist > 0 ? exp(-w) : 1 - w) * wcls[1]; }      /* GetOmega codeml.c:3043-3048 */
         }
         Q[i * n + j] = Q[j * n + i] = q;
      }
   /* times the target's frequency, the diagonal from the row sums — over the elements a row has, in the order of the columns: the same
    * sums, to the bit, as over the whole row (the other elements are zeros) */
   for (i = 0; i < n; i++) {
      double s = 0;
      for (x = cp->nbr_ptr[i]; x < cp->nbr_ptr[i + 1]; x++) { j = cp->nbr[x]; Q[i * n + j] *= pi[j]; s += Q[i * n + j]; }
      Q[i * n + i] = -s;
      mr += pi[i] * s;
   }
   return mr;
}

/* CDF of the omega distribution of the continuous site models M6 (2 gammas), M9 (beta & gamma), M10 (beta & 1 + gamma)
 * (CDFdN_dS codeml.c:2913-2975; parameters in the order of x[]), and the medians of K equal-probability bins inside (1e-7, 99)
 * (DiscreteNSsites codeml.c:2872-2880; the reference's line search to 1e-15 and this bisection find the same roots). */
static double cdf_omega(int nssites, double x, const double *par)
{
   if (nssites == 6) return par[0] * pamlh_gammp(par[1], par[2] * x) + (1 - par[0]) * pamlh_gammp(par[3], par[3] * x);
   if (nssites == 9) return par[0] * (x >= 1 ? 1 : pamlh_betai(par[1], par[2], x)) + (1 - par[0]) * pamlh_gammp(par[3], par[4] * x);
   if (nssites == 10) {
      if (x <= 1) return par[0] * pamlh_betai(par[1], par[2], x);
      return par[0] + (1 - par[0]) * pamlh_gammp(par[3], par[4] * (x - 1));
   }
   if (nssites == 11) {      /* beta & normal truncated to > 1: p0, p, q, mu, s */
      double c;
      if (x <= 1) return par[0] * pamlh_betai(par[1], par[2], x);
      c = pamlh_cdf_normal((par[3] - 1) / par[4]);
      return par[0] + (1 - par[0]) * (1 - pamlh_cdf_normal((par[3] - x) / par[4]) / c);
   }
   if (nssites == 12) {      /* (spike at 0 handled outside) two normals truncated to > 0, means 1 and mu2: p0, p1, mu2, s1, s2 */
      const double f1 = par[1], f2 = 1 - par[1];
      return 1 - f1 * pamlh_cdf_normal(-(x - 1) / par[3]) / pamlh_cdf_normal(1 / par[3])
               - f2 * pamlh_cdf_normal(-(x - par[2]) / par[4]) / pamlh_cdf_normal(par[2] / par[4]);
   }
   {                         /* 13: three normals truncated to > 0, means 0, 1, mu2: p0, p1, mu2, s0, s1, s2 */
      const double f0 = par[0], f1 = par[1], f2 = 1 - f0 - f1;
      return 1 - f0 * 2 * pamlh_cdf_normal(-x / par[3]) - f1 * pamlh_cdf_normal(-(x - 1) / par[4]) / pamlh_cdf_normal(1 / par[4])
               - f2 * pamlh_cdf_normal(-(x - par[2]) / par[5]) / pamlh_cdf_normal(par[2] / par[5]);
   }
}

static void omega_medians(int nssites, const double *par, int K, double *w)
{
   int j, it;
   for (j = 0; j < K; j++) {
      const double pr = (j * 2. + 1) / (2. * K);
      double lo = 1e-7, hi = 99;
      if (cdf_omega(nssites, lo, par) >= pr) { w[j] = lo; continue; }
      if (cdf_omega(nssites, hi, par) <= pr) { w[j] = hi; continue; }
      for (it = 0; it < 200 && hi - lo > 1e-15 * (1 + hi); it++) {
         const double mid = 0.5 * (lo + hi);
         if (cdf_omega(nssites, mid, par) < pr) lo = mid; else hi = mid;
      }
      w[j] = 0.5 * (lo + hi);
   }
}

/* number of exchangeability parameters of a baseml model (nkappa[] baseml.c:1311) */
static int nuc_nkappa(const pamlh *p)
{
   const int m = p->model;
   if (m == REV) return 5;
   if (m == UNREST) return 11;
   if (m == TN93) return p->fix_kappa ? 0 : 2;
   if (m == K80 || m == HKY85 || m == F84 || m == T92) return p->fix_kappa ? 0 : 1;
   return 0;
}

/* UNREST (QUNREST treesub.c:2543): the 11 free off-diagonal rates (row-major, Q[G][A] = 1), pi = the stationary distribution
 * of Q (returned in pi), Q scaled to mean rate 1.  No eigen system: P(t) = matexp(Qt) on the device (PAML_AMD_EIGEN_QMAT). */
static void unrest_set(pamlh *p, int iset, double *pi, const double *rate)
{
   pamlh_eig *e = &p->eig[iset];
   double Q[16], A[5][5], mr = 0;
   int i, j, k = 0, r, c;
   for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) Q[i * 4 + j] = i == j ? 0 : (i * 4 + j == 14 ? 1 : rate[k++]);
   for (i = 0; i < 4; i++) { double t = 0; for (j = 0; j < 4; j++) t += Q[i * 4 + j]; Q[i * 4 + i] = -t; }
   /* pi Q = 0 with sum(pi) = 1: replace the last equation by the normalisation, Gaussian elimination with pivoting */
   for (i = 0; i < 4; i++) { for (j = 0; j < 4; j++) A[i][j] = i < 3 ? Q[j * 4 + i] : 1; A[i][4] = i < 3 ? 0 : 1; }
   for (c = 0; c < 4; c++) {
      int piv = c;
      for (r = c + 1; r < 4; r++) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
      for (j = 0; j < 5; j++) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
      for (r = 0; r < 4; r++)
         if (r != c) { const double f = A[r][c] / A[c][c]; for (j = c; j < 5; j++) A[r][j] -= f * A[c][j]; }
   }
   for (i = 0; i < 4; i++) pi[i] = A[i][4] / A[i][i];
   for (i = 0; i < 4; i++) mr -= pi[i] * Q[i * 4 + i];
   if (!e->U) { e->U = (double *)malloc(16 * 8); e->V = (double *)malloc(16 * 8); e->Root = (double *)malloc(4 * 8); }
   for (i = 0; i < 16; i++) e->U[i] = Q[i] / mr;
   e->kind = PAML_AMD_EIGEN_QMAT;
}

/* eigen system `iset` of a baseml model with frequencies pi and exchangeability parameters kp (those nuc_nkappa counts) */
static void nuc_set(pamlh *p, int iset, const double *pi, const double *kp, double *Q)
{
   const int m = p->model;
   double S[16], mr = 0;
   int i, j, kk;
   pamlh_eig *e = &p->eig[iset];
   if (m == JC69 || m == K80) { e->kind = PAML_AMD_EIGEN_K80; e->kappa = m == JC69 ? 1 : (p->fix_kappa ? p->kappa0 : kp[0]); return; }
   if (m == UNREST) { unrest_set(p, iset, (double *)pi, kp); return; }
   for (i = 0; i < 16; i++) S[i] = 1;
   if (m == HKY85 || m == T92) { const double v = p->fix_kappa ? p->kappa0 : kp[0]; S[0 * 4 + 1] = S[1 * 4 + 0] = S[2 * 4 + 3] = S[3 * 4 + 2] = v; }
   else if (m == F84) {       /* TN93 with kappa1 = 1 + kappa / Y, kappa2 = 1 + kappa / R (QTN93 treesub.c:2179) */
      const double v = p->fix_kappa ? p->kappa0 : kp[0];
      S[0 * 4 + 1] = S[1 * 4 + 0] = 1 + v / (pi[0] + pi[1]); S[2 * 4 + 3] = S[3 * 4 + 2] = 1 + v / (pi[2] + pi[3]);
   }
   else if (m == TN93) {
      const double k1 = p->fix_kappa ? p->kappa0 : kp[0], k2 = p->fix_kappa ? p->kappa0 : kp[1];
      S[0 * 4 + 1] = S[1 * 4 + 0] = k1; S[2 * 4 + 3] = S[3 * 4 + 2] = k2;
   }
   else if (m == REV) {
      S[0 * 4 + 1] = S[1 * 4 + 0] = kp[0]; S[0 * 4 + 2] = S[2 * 4 + 0] = kp[1]; S[0 * 4 + 3] = S[3 * 4 + 0] = kp[2];
      S[1 * 4 + 2] = S[2 * 4 + 1] = kp[3]; S[1 * 4 + 3] = S[3 * 4 + 1] = kp[4];
   }
   for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) Q[i * 4 + j] = (i == j) ? 0 : S[i * 4 + j] * pi[j];
   for (i = 0; i < 4; i++) { double t = 0; for (j = 0; j < 4; j++) t += Q[i * 4 + j]; Q[i * 4 + i] = -t; mr += pi[i] * t; }
   set_eig_uvroot(p, iset, Q, pi, mr);
   if (!e->Cijk) e->Cijk = (double *)malloc(64 * sizeof(double));
   for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) for (kk = 0; kk < 4; kk++) e->Cijk[i * 16 + j * 4 + kk] = e->U[i * 4 + kk] * e->V[kk * 4 + j];
   e->kind = PAML_AMD_EIGEN_CIJK; e->nR = 4;
}

/* nhomo >= 2 (GetInitials baseml.c:1201-1232, SetParameters 1341-1370, GetPMatBranch treesub.c:7503-7519): the branch to node v
 * has its own rate parameters (nhomo 2: kappa; 3, 4 with fix_kappa = 0: the model's kappa / abcde) and, for 3 and 4, the Q of
 * the frequency set of node v; the root's set gives the root distribution.  x after the branch lengths: the rate parameters in
 * tree.branches order, then the frequency sets (3 frequencies each, T92: the GC content) as frequencies (LASTROUND). */
static int nh_nk(const pamlh *p)
{
   static const int nkappa[] = {0, 1, 0, 1, 1, 1, 2, 5, 11};
   return nkappa[p->model];
}
int pamlh_nh_nrate(const pamlh *p) { return p->nhomo == 2 ? p->nbranch : nh_nk(p) * (p->fix_kappa == 0 ? p->nbranch : p->fix_kappa == 1 ? 1 : p->nh_nbtype); }
int pamlh_nh_npi(const pamlh *p)
{
   if (p->nhomo == 5) return p->nh_nbtype + (p->root >= p->ns && p->nh_label[p->root] == p->nh_nbtype);
   return p->nhomo == 4 ? p->nnode : p->nhomo == 3 ? p->ns + 1 + (p->root >= p->ns) : 0;
}
static int nh_piset(const pamlh *p, int v)
{
   if (p->nhomo == 5) return p->nh_label[v];
   return p->nhomo == 4 ? v : (v < p->ns ? v : (v == p->root ? p->ns + 1 : p->ns));
}

static int set_x_nhomo(pamlh *p, const double *x, int *kio, double *Q)
{
   const int nk = nh_nk(p), n31 = p->model == T92 ? 1 : 3, fix = p->fix_kappa;
   const double *rates = x + *kio, *pis = rates + pamlh_nh_nrate(p);
   int i, v;
   for (v = 0; v < p->nnode; v++) {
      double pi[4];
      if (p->nhomo == 2) { for (i = 0; i < 4; i++) pi[i] = p->model == K80 ? 0.25 : p->pi_data[i]; }
      else {
         const double *px = pis + nh_piset(p, v) * n31;
         if (p->model == T92) { pi[0] = pi[2] = (1 - px[0]) / 2; pi[1] = pi[3] = px[0] / 2; }
         else { pi[0] = px[0]; pi[1] = px[1]; pi[2] = px[2]; pi[3] = 1 - px[0] - px[1] - px[2]; }
         if (!(pi[3] > -1e-9)) return pamlh_fail(p, "base frequencies of set %d sum above 1", nh_piset(p, v) + 1);
         /* an estimate on the boundary (the reference prints 0.000000): the symmetrised eigen problem needs pi > 0, P(t) is continuous there */
         for (i = 0; i < 4; i++) if (!(pi[i] > 1e-15)) pi[i] = 1e-15;
      }
      p->eigen_of[v] = v;
      if (v == p->root) {      /* no branch: the set gives com.pi; a placeholder keeps the eigen systems dense */
         memcpy(p->pi, pi, sizeof(pi));
         p->eig[v].kind = PAML_AMD_EIGEN_K80; p->eig[v].kappa = 1;
         continue;
      }
      for (i = 0; i < p->nbranch && p->branch_node[i] != v; i++) ;
      p->fix_kappa = 0;      /* the rate parameters of this branch are always read from x */
      nuc_set(p, v, pi, p->nhomo == 2 ? rates + i : (fix == 0 ? rates + (size_t)i * nk : fix == 1 ? rates : rates + (size_t)p->nh_label[v] * nk), Q);
      p->fix_kappa = fix;
   }
   p->n_labels = p->n_eigen = p->nnode;
   *kio += pamlh_nh_nrate(p) + (p->nhomo > 2 ? pamlh_nh_npi(p) * n31 : 0);
   return 0;
}

/* Several genes (option G).  x = branch lengths, rgene[2..ngene] (rates relative to the first gene, SetParameters
 * baseml.c:1319 / codeml.c:2768), then the substitution parameters — one set (Mgene 0, 2) or one per gene (Mgene 3, 4:
 * SetPGene baseml.c:1428 / codeml.c:2420) — then alpha.  Frequencies: com.pi for all genes (Mgene 0, 3) or com.piG (2, 4).
 * Each gene's Q is scaled to mean rate 1 with its own frequencies and parameters; rgene then multiplies the branch lengths. */
static int set_x_genes(pamlh *p, const double *x, int np, int k, double *Q)
{
   const int n = p->n, G = p->ngene, per_gene = p->mgene >= 3, own_pi = p->mgene == 2 || p->mgene == 4;
   const int nsets = p->mgene >= 2 ? G : 1;
   int g, j, K = 1;
   p->rgene[0] = 1;
   for (g = 1; g < G; g++) p->rgene[g] = x[k++];
   for (g = 0; g < nsets; g++) {
      const double *pi = own_pi ? p->piG[g] : p->pi_data;
      double *pis = p->pi + (size_t)(own_pi ? g : 0) * n;
      memcpy(pis, pi, n * sizeof(double));
      if (p->seqtype == 1) {
         const double *kp = x + k + (per_gene ? g * 2 : 0);
         const double kappa = p->fix_kappa ? p->kappa0 : kp[0], w = p->fix_omega ? p->omega0 : kp[!p->fix_kappa];
         double mr, keep3[12], keep4[4];
         memcpy(keep3, p->fb3x4, sizeof(keep3)); memcpy(keep4, p->fb4, sizeof(keep4));
         if (p->mg && own_pi) { memcpy(p->fb3x4, p->fb3x4G[g], sizeof(keep3)); memcpy(p->fb4, p->fb4G[g], sizeof(keep4)); }      /* F1x4MG / F3x4MG: the gene's own tables (SetPGene codeml.c: com.pf3x4 = com.f3x4[igene]) */
         mr = codon_q_pi(p, pis, kappa, w, Q);
         memcpy(p->fb3x4, keep3, sizeof(keep3)); memcpy(p->fb4, keep4, sizeof(keep4));
         set_eig_uvroot(p, g, Q, pis, mr);
         p->kappa = kappa; p->omega = w; p->class_w[g] = w;
      }
      else if (p->seqtype == 2) {
         double mr = 0;
         int i;
         if (p->aa_model == 0) { for (i = 0; i < 20; i++) pis[i] = 1.0 / 20; p->eig[g].kind = PAML_AMD_EIGEN_JC69LIKE; }
         else {
            if (p->aa_model == 2) memcpy(pis, p->aapi_file, 20 * sizeof(double));
            for (i = 0; i < 20; i++) for (j = 0; j < 20; j++) Q[i * 20 + j] = (i == j) ? 0 : (p->aa_model == 1 ? 1 : p->aaS[i * 20 + j]) * pis[j];      /* model 1: equal exchangeabilities */
            for (i = 0; i < 20; i++) { double t = 0; for (j = 0; j < 20; j++) t += Q[i * 20 + j]; Q[i * 20 + i] = -t; mr += pis[i] * t; }
            set_eig_uvroot(p, g, Q, pis, mr);
         }
      }
      else {
         if (p->model == JC69 || p->model == K80) for (j = 0; j < 4; j++) pis[j] = 0.25;
         nuc_set(p, g, pis, x + k + (per_gene ? g * nuc_nkappa(p) : 0), Q);
      }
   }
   if (p->seqtype == 1) k += (per_gene ? G : 1) * (!p->fix_kappa + !p->fix_omega);
   else if (p->seqtype == 0) k += (per_gene ? G : 1) * nuc_nkappa(p);
   p->n_pi = own_pi ? G : 1;
   p->n_eigen = nsets;
   if (p->malpha) {      /* a gamma shape per gene: class rates [gene][class] (SetPGene(.., _alpha = 1, ..) baseml.c:1460-1463) */
      if (p->ncatG > 60) return pamlh_fail(p, "ncatG too large");
      for (g = 0; g < G; g++) {
         const double alpha = x[k++];
         if (!(alpha > 0)) return pamlh_fail(p, "alpha of gene %d is not positive", g + 1);
         pamlh_discrete_gamma(p->freqK, p->rate + (size_t)g * p->ncatG, alpha, p->ncatG);
         if (g == 0) p->alpha = alpha;
      }
      K = p->ncatG; p->mode = PAML_AMD_MODE_LFUNDG;
   }
   else {
      const double alpha = p->fix_alpha ? p->alpha0 : x[k++];
      p->alpha = alpha;
      if (alpha > 0) {
         if (p->ncatG > 60) return pamlh_fail(p, "ncatG too large");
         pamlh_discrete_gamma(p->freqK, p->rate, alpha, p->ncatG);
         K = p->ncatG; p->mode = PAML_AMD_MODE_LFUNDG;
      }
   }
   p->K = K;
   free(p->gene_eigen_of);
   p->gene_eigen_of = (int *)malloc((size_t)G * K * sizeof(int));
   for (g = 0; g < G; g++) for (j = 0; j < K; j++) p->gene_eigen_of[g * K + j] = p->mgene >= 2 ? g : 0;
   if (k != np) return pamlh_fail(p, "internal: consumed %d of %d parameters", k, np);
   return 0;
}

/* nodes[].branch from the first ntime entries of x (SetBranch treesub.c:3770): the lengths themselves in tree.branches order, the
 * tree file's under fix_blength = 2, or — global clock — age(father) - age(node) from the internal node ages.  -1: not usable. */
int pamlh_x_to_branches(const pamlh *p, const double *x, double *branch)
{
   int i;
   for (i = 0; i < p->nnode; i++) branch[i] = 0;
   if (p->clock) {
      for (i = 0; i < p->nnode; i++) {
         double b;
         if (i == p->root) continue;
         b = x[p->father[i] - p->ns] - (i < p->ns ? (p->tipdate ? p->tip_age[i] : 0) : x[i - p->ns]);
         if (b < -1e-5) return -1;
         branch[i] = b < 0 ? 0 : b;
         if (p->clock == 2 && p->rate_label[i] > 0) branch[i] *= x[p->ns - 1 + p->tipdate + p->rate_label[i] - 1];
         else if (p->tipdate) branch[i] *= x[p->ns - 1];      /* the mutation rate (com.rgene[0] in the reference's likelihood) */
      }
      return 0;
   }
   for (i = 0; i < p->nbranch; i++) {
      const int node = p->branch_node[i];
      if (p->fix_blength == 3) branch[node] = p->tree_branch[node] * x[0];      /* proportional branch lengths (SetBranch treesub.c:3778) */
      else branch[node] = p->ntime ? x[i] : p->tree_branch[node];
      if ((!p->ntime || p->fix_blength == 3) && p->tree_branch[node] < 0) return -1;
   }
   return 0;
}

int pamlh_set_x(pamlh *p, const double *x, int np)
{
   const int n = p->n;
   if (p->ngene > 1 && p->mgene == 1) return pamlh_fail(p, "Mgene = 1: the genes are analysed separately (pamlh_gene_subset)");
   int k = 0, i, j;
   double *Q = (double *)malloc((size_t)n * n * sizeof(double));
   if (np != p->np) { free(Q); return pamlh_fail(p, "expected %d parameters, got %d", p->np, np); }
   /* branch lengths: x[0..ntime) in tree.branches order, or the tree file's when fix_blength = 2 (SetBranch treesub.c:3770) */
   if (pamlh_x_to_branches(p, x, p->branch)) { free(Q); return pamlh_fail(p, p->clock ? "a node is older than its ancestor" : "fix_blength = 2 but the tree has no branch lengths"); }
   k = p->ntime;
   p->n_labels = 1; p->K = 1; p->mode = PAML_AMD_MODE_LFUN; p->n_eigen = 1; p->use_qf = 0; p->n_pi = 1;
   p->freqK[0] = 1; p->rate[0] = 1; p->eigen_of[0] = 0;
   if (p->ngene > 1) { const int rc = set_x_genes(p, x, np, k, Q); free(Q); return rc; }
   if (p->seqtype == 1) {
      double kappa = p->fix_kappa ? p->kappa0 : x[k++];
      memcpy(p->pi, p->pi_data, p->n * sizeof(double));
      p->kappa = kappa;
      if (p->npi) {      /* frequency parameters: com.pi and com.pf3x4 from x (SetParameters codeml.c:2782-2785) */
         for (i = 0; i < ((p->mutsel || p->codonfreq == 1) ? 3 : p->codonfreq == 2 ? 9 : 0); i++)
            if (!(x[k + i] > 0)) { free(Q); return pamlh_fail(p, "frequency ratio %d is not positive", i + 1); }
         codon_freqs_from_x(p, x + k);
         k += p->npi;
      }
      if (p->aadist && p->aadist != 7) {      /* omega a function of the amino-acid distance: x holds a, b */
         const double mr = codon_q_cls(p, p->pi, kappa, 1, x + k, Q);
         if (p->aadist < 0 && !(x[k] <= 1)) { free(Q); return pamlh_fail(p, "aaDist < 0 needs a <= 1"); }
         set_eig_uvroot(p, 0, Q, p->pi, mr);
         p->class_w[0] = x[k + 1];
         k += 2;
      }
      else if (p->aadist == 7) {
         /* AAClasses: omega by the class of the amino-acid pair; with branch labels every label has its own set of class omegas
          * and its own eigen system (SetParameters codeml.c:2804-2812: com.pomega moves on by nOmegaType per label) */
         const int nl = p->model == 2 ? p->n_omega : 1;
         for (j = 0; j < nl; j++) {
            const double mr = codon_q_cls(p, p->pi, kappa, 1, x + k, Q);
            set_eig_uvroot(p, j, Q, p->pi, mr);
            p->eigen_of[j] = j;
            p->class_w[j] = x[k];
            k += p->n_omega_type;
         }
         p->n_eigen = p->n_labels = nl;
      }
      else if (p->nssites == 0 && p->model == 2) {
         /* branch model: label l has its own omega and its own eigen system, each scaled by its own mean rate
          * (SetParameters codeml.c:2804-2812 -> _UU[l]; GetPMatBranch treesub.c:7568-7572) */
         for (j = 0; j < p->n_omega; j++) {
            const double w = (j == p->n_omega - 1 && p->fix_omega) ? p->omega0 : x[k++], mr = codon_q(p, kappa, w, Q);      /* omega_fix: SetParameters codeml.c:2810 */
            set_eig_uvroot(p, j, Q, p->pi, mr);
            p->eigen_of[j] = j;
            p->class_w[j] = w;
         }
         p->n_eigen = p->n_labels = p->n_omega;
      }
      else if (p->model >= 2) {
         /* branch-site models A / B and clade models C / D (SetParametersNSsites codeml.c:2459-2660).  Site classes x branch
          * types pick one of a few omegas:   A/B: back w0 w1 w0 w1, fore w0 w1 w2 w2;   C/D: type b: w0 w1 w_b.
          * The eigen systems are those of the unscaled Q (mr = 1); every branch type has its own time scale
          * Qfactor_NS_branch[b] = 1 / mr(Q at the type's mean omega), applied to t in GetPMatBranch (treesub.c:7547-7554, 7587). */
         const int bs = p->model == 2, nb = p->n_omega, K = bs ? 4 : 3, L = nb, nset = bs ? 3 : 2 + nb;
         double f[4], wc[2], wb[16], t, wsets[18];
         int c, l;
         f[0] = x[k++]; f[1] = x[k++];
         wc[0] = x[k++]; wc[1] = p->nssites == 3 ? x[k++] : 1;
         t = f[0] + f[1];
         if (bs) {
            wb[0] = wb[1] = (p->nssites == 2 && p->fix_omega) ? p->omega0 : x[k++];
            f[2] = t > 1e-100 ? (1 - t) * f[0] / t : -1;
            f[3] = t > 1e-100 ? (1 - t) * f[1] / t : -1;
         }
         else {
            for (l = 0; l < nb; l++) wb[l] = (l == nb - 1 && p->fix_omega) ? p->omega0 : x[k++];
            f[2] = 1 - t;
         }
         wsets[0] = wc[0]; wsets[1] = wc[1];
         if (bs) wsets[2] = wb[1]; else for (l = 0; l < nb; l++) wsets[2 + l] = wb[l];
         for (j = 0; j < nset; j++) { codon_q(p, kappa, wsets[j], Q); set_eig_uvroot(p, j, Q, p->pi, 1.0); }
         for (l = 0; l < L; l++) {
            double wm, qf;
            if (bs) wm = l == 0 ? (t > 1e-100 ? (f[0] * wc[0] + f[1] * wc[1]) / t : 1) : f[0] * wc[0] + f[1] * wc[1] + (1 - t) * wb[1];
            else wm = f[0] * wc[0] + f[1] * wc[1] + f[2] * wb[l];
            qf = 1 / codon_q(p, kappa, wm, Q);
            for (c = 0; c < K; c++) {
               p->qfactor[c * L + l] = qf;
               p->eigen_of[c * L + l] = bs ? (l == 0 ? c % 2 : (c <= 1 ? c : 2)) : (c < 2 ? c : 2 + l);
            }
         }
         for (c = 0; c < K; c++) { p->freqK[c] = f[c]; p->rate[c] = 1; p->class_w[c] = c < 2 ? wc[c] : wb[nb - 1]; }
         p->K = K; p->n_eigen = nset; p->n_labels = L; p->mode = PAML_AMD_MODE_LFUNDG; p->use_qf = 1;
      }
      else if (p->nssites == 0) {
         double w = p->fix_omega ? p->omega0 : x[k++], mr = codon_q(p, kappa, w, Q);
         p->omega = w;
         set_eig_uvroot(p, 0, Q, p->pi, mr);
      }
      else {
         double w[16], f[16], mr, wmean = 0;
         int K;
         if (p->nssites == 1) { f[0] = x[k]; w[0] = x[k + 1]; f[1] = 1 - f[0]; w[1] = 1; K = 2; k += 2; }
         else if (p->nssites == 2) { f[0] = x[k]; f[1] = x[k + 1]; f[2] = 1 - f[0] - f[1]; w[0] = x[k + 2]; w[1] = 1; w[2] = p->fix_omega ? p->omega0 : x[k + 3]; K = 3; k += 3 + !p->fix_omega; }
         else if (p->nssites == 5) {      /* M5 (gamma): medians of K equal-probability bins of gamma(a, b), kept inside (1e-7, 99) (DiscreteNSsites codeml.c:2873-2880) */
            K = p->ncatG;
            if (K > 16) { free(Q); return pamlh_fail(p, "ncatG too large"); }
            for (j = 0; j < K; j++) {
               w[j] = pamlh_quantile_gamma((j * 2. + 1) / (2. * K), x[k], x[k + 1]);
               w[j] = w[j] < 1e-7 ? 1e-7 : w[j] > 99 ? 99 : w[j];
               f[j] = 1.0 / K;
            }
            k += 2;
         }
         else if (p->nssites == 6 || (p->nssites >= 9 && p->nssites <= 13)) {
            const int npar = p->nssites == 6 ? 4 : p->nssites == 13 ? 6 : 5;
            K = p->ncatG;
            if (K > 16) { free(Q); return pamlh_fail(p, "ncatG too large"); }
            if (p->nssites == 12) {      /* a spike at omega = 0 (proportion p0) + ncatG classes of the normal mixture, as M8 adds its class to
                                          * the ncatG of the control file (codeml.c:2888-2895) */
               if (K > 15) { free(Q); return pamlh_fail(p, "ncatG too large"); }
               omega_medians(12, x + k, K, w + 1);
               w[0] = 0; f[0] = x[k];
               for (j = 1; j <= K; j++) f[j] = (1 - x[k]) / K;
               K++;
            }
            else {
               omega_medians(p->nssites, x + k, K, w);
               for (j = 0; j < K; j++) f[j] = 1.0 / K;
            }
            k += npar;
         }
         else if (p->nssites == 4) {      /* M4 (freqs, NSfreqs codeml.c:2531-2538) */
            static const double w4[5] = {0, 1. / 3, 2. / 3, 1, 3};
            K = 5;
            for (j = 0, f[4] = 1; j < 4; j++) f[4] -= (f[j] = x[k++]);
            for (j = 0; j < 5; j++) w[j] = w4[j];
         }
         else if (p->nssites == 3) {      /* M3 (discrete): K-1 proportions then K omegas */
            K = p->ncatG;
            for (j = 0, f[K - 1] = 1; j < K - 1; j++) f[K - 1] -= (f[j] = x[k++]);
            for (j = 0; j < K; j++) w[j] = x[k++];
         }
         else {   /* M7 / M8: K = ncatG median quantiles of beta(p, q) (DiscreteNSsites codeml.c:2869-2874) */
            const int off = p->nssites == 8;
            const double bp = x[k + off], bq = x[k + off + 1];
            K = p->ncatG;
            if (K + off > 15) { free(Q); return pamlh_fail(p, "ncatG too large"); }
            for (j = 0; j < K; j++) { w[j] = pamlh_quantile_beta((j * 2. + 1) / (2. * K), bp, bq); f[j] = 1.0 / K; }
            if (off) {
               const double p0 = x[k];
               for (j = 0; j < K; j++) f[j] *= p0;
               f[K] = 1 - p0;
               w[K] = p->fix_omega ? p->omega0 : x[k + 3];
               k += 3 + !p->fix_omega;
               K++;
            }
            else k += 2;
         }
         /* Qfactor_NS = 1 / mr(Q at the mean omega) (codeml.c:2586-2605); class ir: Root /= 1/Qfactor_NS (treesub.c:7675-7685) */
         for (j = 0; j < K; j++) wmean += f[j] * w[j];
         mr = codon_q(p, kappa, wmean, Q);
         p->ns_mr = mr;
         for (j = 0; j < K; j++) {
            codon_q(p, kappa, w[j], Q);
            set_eig_uvroot(p, j, Q, p->pi, mr);
            p->freqK[j] = f[j]; p->rate[j] = 1; p->eigen_of[j] = j; p->class_w[j] = w[j];
         }
         p->K = K; p->n_eigen = K; p->mode = PAML_AMD_MODE_LFUNDG;
      }
   }
   else if (p->seqtype == 2) {
      double mr = 0;
      if (p->aa_model == 0) {
         for (i = 0; i < 20; i++) p->pi[i] = 1.0 / 20;
         p->eig[0].kind = PAML_AMD_EIGEN_JC69LIKE;
      }
      else if (p->aa_model >= 8) {
         double S[400] = {0};
         memcpy(p->pi, p->pi_data, 20 * sizeof(double));
         for (i = 1; i < 20; i++) for (j = 0; j < i; j++) if (p->aa1step[i * 20 + j]) S[i * 20 + j] = S[j * 20 + i] = x[k++];
         S[19 * 20 + 9] = S[9 * 20 + 19] = 1;
         for (i = 0; i < 20; i++) for (j = 0; j < 20; j++) Q[i * 20 + j] = (i == j) ? 0 : S[i * 20 + j] * p->pi[j];
         for (i = 0; i < 20; i++) { double sm = 0; for (j = 0; j < 20; j++) sm += Q[i * 20 + j]; Q[i * 20 + i] = -sm; mr += p->pi[i] * sm; }
         set_eig_uvroot(p, 0, Q, p->pi, mr);
      }
      else if (p->aa_model == 6) {
         /* FromCodon: exchangeability of two amino acids = the codon chain's flow between their codon sets, under codon
          * frequencies fb61 and kappa, over the product of the two amino-acid frequencies (Qcodon2aa codeml.c:3487-3523;
          * eigenQaa 3418-3422, 3441-3447); omega cancels in the scaling */
         const double kappa = p->fix_kappa ? p->kappa0 : x[k++];
         double S[400] = {0}, piaa[20] = {0};
         int from61[64], nc = 0, c, a, b;
         memcpy(p->pi, p->pi_data, 20 * sizeof(double));
         for (c = 0; c < 64; c++) if (p->code[c] != '*') from61[nc++] = c;
         for (c = 0; c < nc; c++) piaa[aa_of_codon(p, from61[c])] += p->fb61[c];
         for (i = 0; i < nc; i++) {
            a = aa_of_codon(p, from61[i]);
            for (j = 0; j < i; j++) {
               const int c1 = from61[i], c2 = from61[j];
               const int f[3] = {c1 / 16, (c1 / 4) % 4, c1 % 4}, t[3] = {c2 / 16, (c2 / 4) % 4, c2 % 4};
               int nd = 0, pos = 0, q;
               double v;
               b = aa_of_codon(p, c2);
               for (q = 0; q < 3; q++) if (f[q] != t[q]) { nd++; pos = q; }
               if (nd != 1 || a == b || piaa[a] == 0 || piaa[b] == 0) continue;
               v = (f[pos] + t[pos] == 1 || f[pos] + t[pos] == 5) ? kappa : 1;
               v *= p->fb61[i] / piaa[a] * p->fb61[j] / piaa[b];
               S[a * 20 + b] += v; S[b * 20 + a] += v;
            }
         }
         for (i = 0; i < 20; i++) for (j = 0; j < 20; j++) Q[i * 20 + j] = (i == j) ? 0 : S[i * 20 + j] * p->pi[j];
         for (i = 0; i < 20; i++) { double s = 0; for (j = 0; j < 20; j++) s += Q[i * 20 + j]; Q[i * 20 + i] = -s; mr += p->pi[i] * s; }
         set_eig_uvroot(p, 0, Q, p->pi, mr);
         p->kappa = kappa;
      }
      else {
         memcpy(p->pi, p->aa_model == 2 ? p->aapi_file : p->pi_data, 20 * sizeof(double));   /* model 2: file pi, used as read */
         for (i = 0; i < 20; i++) for (j = 0; j < 20; j++) Q[i * 20 + j] = (i == j) ? 0 : (p->aa_model == 1 ? 1 : p->aaS[i * 20 + j]) * p->pi[j];   /* model 1 (EqualInput): rates proportional to the target frequency */
         for (i = 0; i < 20; i++) { double s = 0; for (j = 0; j < 20; j++) s += Q[i * 20 + j]; Q[i * 20 + i] = -s; mr += p->pi[i] * s; }
         set_eig_uvroot(p, 0, Q, p->pi, mr);
      }
   }
   else {
      const int m = p->model;
      double S[16], mr = 0;
      for (i = 0; i < 16; i++) S[i] = 1;
      if (m == JC69 || m == K80) for (i = 0; i < 4; i++) p->pi[i] = 0.25;
      else memcpy(p->pi, p->pi_data, 4 * sizeof(double));
      if (p->nhomo >= 2) { const int rc = set_x_nhomo(p, x, &k, Q); if (rc) { free(Q); return rc; } }
      else {
      if (p->nhomo == 1) {       /* base frequencies are parameters, after the rate parameters in x (SetParameters baseml.c:1328-1341) */
         const double *px = x + k + nuc_nkappa(p);
         if (m == T92) { p->pi[0] = p->pi[2] = (1 - px[0]) / 2; p->pi[1] = p->pi[3] = px[0] / 2; }
         else { p->pi[0] = px[0]; p->pi[1] = px[1]; p->pi[2] = px[2]; p->pi[3] = 1 - px[0] - px[1] - px[2]; }
         if (!(p->pi[3] > 1e-9)) { free(Q); return pamlh_fail(p, "base frequencies sum above 1"); }
      }
      if (m == UNREST) { unrest_set(p, 0, p->pi, x + k); k += 11; }
      else if (m == JC69 || m == K80) {
         p->eig[0].kind = PAML_AMD_EIGEN_K80;
         p->eig[0].kappa = m == JC69 ? 1 : (p->fix_kappa ? p->kappa0 : x[k++]);
      }
      else {
         if (m == HKY85 || m == T92) { double kp = p->fix_kappa ? p->kappa0 : x[k++]; S[0 * 4 + 1] = S[1 * 4 + 0] = S[2 * 4 + 3] = S[3 * 4 + 2] = kp; }
         else if (m == F84) {
            double kp = p->fix_kappa ? p->kappa0 : x[k++];
            S[0 * 4 + 1] = S[1 * 4 + 0] = 1 + kp / (p->pi[0] + p->pi[1]); S[2 * 4 + 3] = S[3 * 4 + 2] = 1 + kp / (p->pi[2] + p->pi[3]);
         }
         else if (m == TN93) {
            double k1 = p->fix_kappa ? p->kappa0 : x[k], k2 = p->fix_kappa ? p->kappa0 : x[k + 1];
            if (!p->fix_kappa) k += 2;
            S[0 * 4 + 1] = S[1 * 4 + 0] = k1; S[2 * 4 + 3] = S[3 * 4 + 2] = k2;
         }
         else if (m == REV) {   /* TC, TA, TG, CA, CG relative to AG = 1 (treesub.c:2499-2505) */
            S[0 * 4 + 1] = S[1 * 4 + 0] = x[k]; S[0 * 4 + 2] = S[2 * 4 + 0] = x[k + 1]; S[0 * 4 + 3] = S[3 * 4 + 0] = x[k + 2];
            S[1 * 4 + 2] = S[2 * 4 + 1] = x[k + 3]; S[1 * 4 + 3] = S[3 * 4 + 1] = x[k + 4];
            k += 5;
         }
         for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) Q[i * 4 + j] = (i == j) ? 0 : S[i * 4 + j] * p->pi[j];
         for (i = 0; i < 4; i++) { double s = 0; for (j = 0; j < 4; j++) s += Q[i * 4 + j]; Q[i * 4 + i] = -s; mr += p->pi[i] * s; }
         set_eig_uvroot(p, 0, Q, p->pi, mr);
         {  /* baseml's P(t) goes through Cijk (PMatCijk baseml.c:1572): fold U, V into Cijk[i][j][k], nR = 4 */
            pamlh_eig *e = &p->eig[0];
            int kk;
            if (!e->Cijk) e->Cijk = (double *)malloc(64 * sizeof(double));
            for (i = 0; i < 4; i++) for (j = 0; j < 4; j++) for (kk = 0; kk < 4; kk++) e->Cijk[i * 16 + j * 4 + kk] = e->U[i * 4 + kk] * e->V[kk * 4 + j];
            e->kind = PAML_AMD_EIGEN_CIJK; e->nR = 4;
         }
      }
      }
   }
   if (p->seqtype == 0 && p->nhomo == 1) k += p->model == T92 ? 1 : 3;
   /* gamma rates for sites (not with NSsites): alpha fixed > 0 or free */
   if (!(p->seqtype == 1 && p->nssites)) {
      double alpha = p->fix_alpha ? p->alpha0 : x[k++];
      p->alpha = alpha;
      if (alpha > 0) {
         if (p->ncatG > 60) { free(Q); return pamlh_fail(p, "ncatG too large"); }
         pamlh_discrete_gamma(p->freqK, p->rate, alpha, p->ncatG);
         p->K = p->ncatG; p->mode = PAML_AMD_MODE_LFUNDG;
         for (j = 0; j < p->K; j++) { int l; for (l = 0; l < p->n_labels; l++) p->eigen_of[j * p->n_labels + l] = p->n_labels > 1 ? l : 0; }
      }
      p->adg = 0;
      if (!p->fix_rho || p->rho0 != 0) {      /* AutodGamma: MK and the same class rates (SetParameters baseml.c:1388-1391) */
         p->rho = p->fix_rho ? p->rho0 : x[k++];
         if (alpha > 0 && p->rho != 0) {
            if (p->ncatG > 32) { free(Q); return pamlh_fail(p, "ncatG too large for the auto-discrete-gamma model"); }
            pamlh_autod_gamma(p->MK, p->freqK, p->rate, alpha, p->rho, p->ncatG);
            p->adg = 1;
         }
      }
   }
   free(Q);
   if (k != np) return pamlh_fail(p, "internal: consumed %d of %d parameters", k, np);
   return 0;
}

int pamlh_mgene(const pamlh *p) { return p->ngene > 1 ? p->mgene : 0; }
int pamlh_malpha(const pamlh *p) { return p->malpha; }

/* Mgene = 1: gene g of the data set as an analysis of its own — its patterns, weights and site map, frequencies counted from
 * its sites, the same tree and options, one gene, its own parameter vector (MultipleGenes / GetSubSeqs in the reference).
 * Free with pamlh_free. */
int pamlh_gene_subset(const pamlh *p, int g, pamlh **out)
{
   pamlh *q;
   int i, j, h0, h1, np1, nsite = 0;
   *out = NULL;
   if (g < 0 || g >= p->ngene) return -1;
   h0 = p->posG[g]; h1 = p->posG[g + 1]; np1 = h1 - h0;
   q = (pamlh *)malloc(sizeof(pamlh));
   *q = *p;
   q->eng = NULL; q->err[0] = 0; q->gene_eigen_of = NULL;
   q->ngene = 1; q->mgene = 0; q->npatt = np1; q->posG[0] = 0; q->posG[1] = np1;
   q->names = (char **)calloc(p->ns, sizeof(char *));
   for (i = 0; i < p->ns; i++) { q->names[i] = (char *)malloc(strlen(p->names[i]) + 1); strcpy(q->names[i], p->names[i]); }
   q->z = (unsigned char *)malloc((size_t)p->ns * np1);
   q->raw = (char *)malloc((size_t)p->ns * np1 * p->n31);
   for (i = 0; i < p->ns; i++) {
      memcpy(q->z + (size_t)i * np1, p->z + (size_t)i * p->npatt + h0, np1);
      memcpy(q->raw + (size_t)i * np1 * p->n31, p->raw + ((size_t)i * p->npatt + h0) * p->n31, (size_t)np1 * p->n31);
   }
   q->w = (double *)malloc(np1 * sizeof(double));
   memcpy(q->w, p->w + h0, np1 * sizeof(double));
   q->pose = (int *)malloc((p->n_pose + 1) * sizeof(int));
   for (i = 0; i < p->n_pose; i++) if (p->pose[i] >= h0 && p->pose[i] < h1) q->pose[nsite++] = p->pose[i] - h0;
   q->n_pose = nsite; q->ls = nsite; q->lgene[0] = nsite;
   q->n_chara = (int *)malloc(p->n_codes * sizeof(int)); memcpy(q->n_chara, p->n_chara, p->n_codes * sizeof(int));
   q->chara_map = (unsigned char *)malloc((size_t)p->n_codes * p->n); memcpy(q->chara_map, p->chara_map, (size_t)p->n_codes * p->n);
#define DUP(field, count, type) do { q->field = (type *)malloc((size_t)(count) * sizeof(type)); memcpy(q->field, p->field, (size_t)(count) * sizeof(type)); } while (0)
   DUP(sons_ptr, p->nnode + 1, int); DUP(sons, p->sons_ptr[p->nnode], int); DUP(label, p->nnode, int); DUP(branch_node, 2 * p->ns, int);
   DUP(father, 2 * p->ns, int); DUP(tree_branch, 2 * p->ns, double);
   if (p->scale) DUP(scale, p->nnode, unsigned char);
   if (p->rate_label) DUP(rate_label, p->nnode, int);
   if (p->nh_label) DUP(nh_label, p->nnode, int);
   if (p->tip_age) { DUP(tip_age, p->nnode, double); DUP(age_low, p->nnode, double); }
#undef DUP
   q->branch = (double *)calloc(p->nnode, sizeof(double));
   q->pi = (double *)calloc(64 * PAMLH_MAXGENE, sizeof(double));
   q->freqK = (double *)calloc(64, sizeof(double));
   q->rate = (double *)calloc(64 * PAMLH_MAXGENE, sizeof(double));
   q->eigen_of = (int *)calloc(64 * PAMLH_MAXEIG, sizeof(int));
   for (i = 0; i < PAMLH_MAXEIG; i++) { q->eig[i].U = q->eig[i].V = q->eig[i].Root = q->eig[i].Cijk = q->eig[i].Q = q->eig[i].qpi = q->eig[i].qv = NULL; q->eig[i].lazy = q->eig[i].qv_n = 0; }
   /* frequencies of this gene alone, then the one-gene parameter count */
   if (q->seqtype == 1) freqs_codon(q); else freqs_base_aa(q);
   if (q->seqtype == 0 && q->model == T92) { q->pi_data[0] = q->pi_data[2] = (q->pi_data[0] + q->pi_data[2]) / 2; q->pi_data[1] = q->pi_data[3] = (q->pi_data[1] + q->pi_data[3]) / 2; }
   q->np = p->np - (p->ngene - 1);      /* no rgene */
   j = 0; (void)j;
   *out = q;
   return 0;
}

/* A private copy of the model state (what pamlh_set_x writes: branch lengths, pi, class tables, eigen systems) that shares the
 * data, tree and options read-only with `p`: concurrent pamlh_set_x calls on different copies do not interfere. */
pamlh *pamlh_state_clone(const pamlh *p)
{
   pamlh *q = (pamlh *)malloc(sizeof(pamlh));
   int i;
   if (!q) return NULL;
   *q = *p;
   q->eng = NULL;
   q->err[0] = 0;
   q->branch = (double *)calloc(p->nnode, sizeof(double));
   q->pi = (double *)calloc(64 * PAMLH_MAXGENE, sizeof(double));
   q->gene_eigen_of = NULL;
   q->freqK = (double *)calloc(64, sizeof(double));
   q->rate = (double *)calloc(64 * PAMLH_MAXGENE, sizeof(double));
   q->eigen_of = (int *)calloc(64 * PAMLH_MAXEIG, sizeof(int));
   for (i = 0; i < PAMLH_MAXEIG; i++) { q->eig[i].U = q->eig[i].V = q->eig[i].Root = q->eig[i].Cijk = q->eig[i].Q = q->eig[i].qpi = q->eig[i].qv = NULL; q->eig[i].lazy = q->eig[i].qv_n = 0; }
   return q;
}

void pamlh_state_free(pamlh *q)
{
   int i;
   if (!q) return;
   for (i = 0; i < PAMLH_MAXEIG; i++) pamlh_eig_release(&q->eig[i]);
   free(q->branch); free(q->pi); free(q->freqK); free(q->rate); free(q->eigen_of); free(q->gene_eigen_of);
   free(q);
}

/* the engine for this data set and tree (created on first use) */
/* Multi-GPU (one process per GPU): keep only this rank's contiguous block of site patterns (paml_amd_shard_bounds) and remember
 * the communicator's id; the engine joins it when it is created, and from then on every evaluation returns the total over the
 * ranks (paml_amd_comm_init) — every rank sees the same lnL bits, so they all take the same optimisation steps.  Call after
 * pamlh_load (frequencies were estimated from the whole alignment there) and before the first evaluation.  Per-site outputs (lnf,
 * NEB / BEB, ancestral states) are not available on a shard.  id128: the 128 bytes of paml_amd_comm_unique_id from rank 0
 * (NULL with world = 1: no communicator). */
int pamlh_set_shard(pamlh *p, int rank, int world, const void *id128)
{
   long first = 0, count = 0;
   int i;
   unsigned char *z;
   if (p->eng) return pamlh_fail(p, "set_shard: call before the first evaluation");
   /* (refused for every rank alike — the answer depends on npatt and world only — so the ranks of a job fail together, before any collective call) */
   if (paml_amd_shard_bounds(p->npatt, world, rank, &first, &count) || count < 1)
      return pamlh_fail(p, "set_shard: %d site patterns can be sharded over at most %d GPUs (asked for %d)", p->npatt, paml_amd_max_ranks(p->npatt), world);
   z = (unsigned char *)malloc((size_t)p->ns * count);
   for (i = 0; i < p->ns; i++) memcpy(z + (size_t)i * count, p->z + (size_t)i * p->npatt + first, count);
   memmove(p->w, p->w + first, count * sizeof(double));
   free(p->z);
   p->z = z;
   p->npatt_global = p->npatt; p->shard_first = first; p->shard_rank = rank; p->shard_world = world;
   p->npatt = (int)count;
   p->shard_have_id = id128 != NULL;
   if (id128) memcpy(p->shard_id, id128, PAML_AMD_COMM_ID_BYTES);
   /* gene boundaries stay where they are in the global range: a shard holds the part of each gene inside it, possibly nothing */
   for (i = 0; i <= p->ngene; i++) { long b = (long)p->posG[i] - first; p->posG[i] = (int)(b < 0 ? 0 : b > count ? count : b); }
   return 0;
}

int pamlh_engine_ready(pamlh *p)
{
   int rc;
   if (p->eng) return 0;
   if ((rc = paml_amd_create(&p->eng, p->n, p->ns, p->npatt, 64, p->ngene, (p->shard_world > 0 && p->npatt_global != p->npatt) ? PAML_AMD_SHARD : 0)))
      return pamlh_fail(p, "paml_amd_create failed (%d): no GPU?", rc);
   if ((rc = paml_amd_set_tips(p->eng, p->z, p->cleandata, p->n_codes, p->n_chara, p->chara_map, p->w, p->ngene > 1 ? p->posG : NULL)) ||
       (rc = paml_amd_set_tree(p->eng, p->nnode, p->root, p->sons_ptr, p->sons, p->label, p->scale)))
      return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   if (p->shard_world > 0 &&
       (rc = paml_amd_comm_init(p->eng, p->shard_rank, p->shard_world, p->shard_have_id ? p->shard_id : NULL, p->npatt_global, p->shard_first)))
      return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   return 0;
}

/* class proportions of the current model state are a probability vector */
int pamlh_model_feasible(const pamlh *p)
{
   int i;
   for (i = 0; i < p->K; i++)
      if (!(p->freqK[i] >= 0)) return 0;
   return 1;
}

/* `method` of the control file (0: all parameters at once, 1: one branch at a time) */
int pamlh_n_trees(const pamlh *p) { return p->ntrees; }
const char *pamlh_ctl_option(const pamlh *p, const char *key) { return pamlh_opt(p, key); }
int pamlh_method(const pamlh *p) { return (int)pamlh_optd(p, "method", 0); }

/* the engine behind this analysis (NULL before the first evaluation): counters, profiling */
void *pamlh_engine_handle(const pamlh *p) { return p ? (void *)p->eng : NULL; }

int pamlh_engine_model(pamlh *p)
{
   int rc;
   if ((rc = pamlh_engine_ready(p))) return rc;
   if ((rc = paml_amd_set_pi(p->eng, p->n_pi, p->pi))) return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   if ((rc = pamlh_upload_eigen_sets(p, p->eng, 0, NULL))) return rc;
   if ((rc = paml_amd_set_classes(p->eng, p->mode, p->K, p->freqK, p->rate, p->n_labels, p->ngene > 1 ? p->gene_eigen_of : p->eigen_of,
                                  p->use_qf ? p->qfactor : NULL)) ||
       (p->malpha && (rc = paml_amd_set_gene_class_rates(p->eng, p->rate))))
      return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   return 0;
}

int pamlh_eval_gpu(pamlh *p, double *lnL, double *lnf)
{
   int i, rc;
   if ((rc = pamlh_engine_ready(p))) return rc;
   if ((rc = paml_amd_set_pi(p->eng, p->n_pi, p->pi))) return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   if ((rc = pamlh_upload_eigen_sets(p, p->eng, 0, NULL))) return rc;
   if ((rc = paml_amd_set_classes(p->eng, p->mode, p->K, p->freqK, p->rate, p->n_labels, p->ngene > 1 ? p->gene_eigen_of : p->eigen_of,
                                  p->use_qf ? p->qfactor : NULL)) ||
       (p->malpha && (rc = paml_amd_set_gene_class_rates(p->eng, p->rate))))
      return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   if (p->adg) {      /* lfunAdG: fx_r on the device, the rate chain over the sites in their original order on the host */
      if (lnf) for (i = 0; i < p->npatt; i++) lnf[i] = 0;      /* sites are not independent: no per-pattern log f */
      rc = paml_amd_eval_adg(p->eng, p->branch, NULL, p->MK, p->pose, p->n_pose, lnL);
   }
   else rc = paml_amd_eval(p->eng, p->branch, p->ngene > 1 ? p->rgene : NULL, lnL, lnf, NULL);
   if (rc == PAML_AMD_ENOCONV && pamlh_force_host_eigen()) {      /* the matrices are still here (pamlh_eig::Q): host decomposition, once more */
      for (i = 0; i < p->n_eigen; i++) pamlh_eig_host(&p->eig[i], p->n);
      return pamlh_eval_gpu(p, lnL, lnf);
   }
   if (rc) return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   return 0;
}

/* The exact-likelihood consumer of mcmctree (usedata = 1): lnpD_locus (mcmctree.c:1130-1166) turns the sampled node ages and
 * rates into the gene tree's branch lengths — global clock: (age of the father - age of the node) x rgene[locus]; independent /
 * correlated rates: (age difference) x the branch's own rate (every species sampled once: the species-tree path of a gene-tree
 * branch is that branch) — and calls com.plfun(NULL, -1): one likelihood evaluation at those lengths, the substitution model
 * untouched.  Here: age[nnode] (tips too: 0, or their sampling dates), rate = NULL with rgene for the clock, or rate[nnode] per
 * branch.  The model state is the last pamlh_set_x (kappa, alpha ...); model_changed = 0 skips re-sending the eigen systems —
 * an MCMC proposal that moves ages or rates only costs the branch lengths (232 bytes) and one evaluation. */
int pamlh_lnpd_locus(pamlh *p, const double *age, double rgene, const double *rate, int model_changed, double *lnL)
{
   int i, rc;
   double *br;
   if (!p || !age || !lnL) return -1;
   if (p->adg) return pamlh_fail(p, "lnpd_locus: not with rho");
   if (model_changed || !p->eng) { if ((rc = pamlh_engine_model(p))) return rc; }
   br = (double *)malloc(p->nnode * sizeof(double));
   for (i = 0; i < p->nnode; i++) {
      br[i] = 0;
      if (i == p->root) continue;
      br[i] = (age[p->father[i]] - age[i]) * (rate ? rate[i] : rgene);
      if (br[i] < 0) { free(br); return pamlh_fail(p, "lnpd_locus: node %d is older than its father (blength < 0, mcmctree.c:1150)", i + 1); }
   }
   rc = paml_amd_eval(p->eng, br, p->ngene > 1 ? p->rgene : NULL, lnL, NULL, NULL);
   free(br);
   return rc ? pamlh_fail(p, "%s", paml_amd_last_error(p->eng)) : 0;
}

/* Name of parameter i of x[] (branch lengths are "t <node>..<node>" in the reference's numbering, tree.branches order). */
int pamlh_param_name(const pamlh *p, int i, char *buf, int cap)
{
   int k = p->ntime, g, j;
   const int rep = (p->ngene > 1 && p->mgene >= 3) ? p->ngene : 1;
   if (i < 0 || i >= p->np) return -1;
   if (i < p->ntime && p->fix_blength == 3) { snprintf(buf, cap, "branch-length scale"); return 0; }
   if (i < p->ns - 1 && p->clock) { snprintf(buf, cap, "age of node %d", p->ns + i + 1); return 0; }
   if (i == p->ns - 1 && p->clock && p->tipdate) { snprintf(buf, cap, "mutation rate per time unit"); return 0; }
   if (i < p->ntime && p->clock) { snprintf(buf, cap, "rate of branch class %d", i - (p->ns - 1) - p->tipdate + 1); return 0; }
   if (i < p->ntime) { const int node = p->branch_node[i]; snprintf(buf, cap, "t %d..%d", p->father[node] + 1, node + 1); return 0; }
   if (i < k + p->ngene - 1) { snprintf(buf, cap, "rgene%d", i - k + 2); return 0; }
   k += p->ngene - 1;
#define NAME(...) do { if (i == k) { snprintf(buf, cap, __VA_ARGS__); return 0; } k++; } while (0)
   for (g = 0; g < rep; g++) {
      char sfx[16] = "";
      if (rep > 1) snprintf(sfx, sizeof(sfx), " (gene %d)", g + 1);
      if (p->seqtype == 1) {
         if (!p->fix_kappa) NAME("kappa%s", sfx);
         for (j = 0; j < p->npi; j++) NAME("codon frequency parameter %d", j + 1);
         if (p->aadist == 7) { int l; for (l = 0; l < (p->model == 2 ? p->n_omega : 1); l++) for (j = 0; j < p->n_omega_type; j++) NAME("omega class %d (branch type %d)", j, l); }
         else if (p->aadist) { NAME("a (omega against amino-acid distance)"); NAME("b"); }
         else if (p->nssites == 0 && p->model == 2) { for (j = 0; j < p->n_omega - (p->fix_omega != 0); j++) NAME("omega #%d", j); }
         else if (p->model >= 2) {
            NAME("p0"); NAME("p1"); NAME("w0");
            if (p->nssites == 3) NAME("w1");
            if (p->model == 2) { if (p->nssites == 3 || !p->fix_omega) NAME("w2 (foreground)"); }
            else for (j = 0; j < p->n_omega - (p->fix_omega != 0); j++) NAME("w%d (branch type %d)", 2 + j, j);
         }
         else if (p->nssites == 0) { if (!p->fix_omega) NAME("omega%s", sfx); }
         else if (p->nssites == 1) { NAME("p0"); NAME("w0"); }
         else if (p->nssites == 2) { NAME("p0"); NAME("p1"); NAME("w0"); if (!p->fix_omega) NAME("w2"); }
         else if (p->nssites == 3) { for (j = 0; j < p->ncatG - 1; j++) NAME("p%d", j); for (j = 0; j < p->ncatG; j++) NAME("w%d", j); }
         else if (p->nssites == 4) { for (j = 0; j < 4; j++) NAME("p%d", j); }
         else if (p->nssites == 5) { NAME("a (gamma)"); NAME("b (gamma)"); }
         else if (p->nssites == 6) { NAME("p0"); NAME("a1"); NAME("b1"); NAME("a2"); }
         else if (p->nssites == 9 || p->nssites == 10) { NAME("p0"); NAME("p (beta)"); NAME("q (beta)"); NAME("a (gamma)"); NAME("b (gamma)"); }
         else if (p->nssites == 11) { NAME("p0"); NAME("p (beta)"); NAME("q (beta)"); NAME("mu"); NAME("s"); }
         else if (p->nssites == 12) { NAME("p0"); NAME("p1"); NAME("mu2"); NAME("s1"); NAME("s2"); }
         else if (p->nssites == 13) { NAME("p0"); NAME("p1"); NAME("mu2"); NAME("s0"); NAME("s1"); NAME("s2"); }
         else if (p->nssites == 7) { NAME("p (beta)"); NAME("q (beta)"); }
         else if (p->nssites == 8) { NAME("p0"); NAME("p (beta)"); NAME("q (beta)"); if (!p->fix_omega) NAME("ws"); }
      }
      else if (p->seqtype == 2) {
         if (p->aa_model == 6 && !p->fix_kappa) NAME("kappa");
         if (p->aa_model >= 8) { static const char AAS[] = "ARNDCQEGHILKMFPSTWYV"; int a, b; for (a = 1; a < 20; a++) for (b = 0; b < a; b++) if (p->aa1step[a * 20 + b]) NAME("exchangeability %c-%c", AAS[a], AAS[b]); }
      }
      else if (p->seqtype == 0 && p->nhomo < 2) {
         if (p->model == UNREST) { for (j = 0; j < 11; j++) NAME("rate %d%s", j + 1, sfx); }
         else if (p->model == REV) { static const char *const r[5] = {"a (TC)", "b (TA)", "c (TG)", "d (CA)", "e (CG)"}; for (j = 0; j < 5; j++) NAME("%s%s", r[j], sfx); }
         else if (p->model == TN93 && !p->fix_kappa) { NAME("kappa1%s", sfx); NAME("kappa2%s", sfx); }
         else if ((p->model == K80 || p->model == HKY85 || p->model == F84 || p->model == T92) && !p->fix_kappa) NAME("kappa%s", sfx);
      }
   }
   if (p->seqtype == 0 && p->nhomo == 1) { if (p->model == T92) NAME("GC content"); else { NAME("pi_T"); NAME("pi_C"); NAME("pi_A"); } }
   if (p->seqtype == 0 && p->nhomo >= 2) {
      for (j = 0; j < pamlh_nh_nrate(p); j++) NAME("rate parameter %d", j + 1);
      for (j = 0; j < (p->nhomo > 2 ? pamlh_nh_npi(p) : 0); j++) { if (p->model == T92) NAME("GC content (set %d)", j + 1); else { NAME("pi_T (set %d)", j + 1); NAME("pi_C (set %d)", j + 1); NAME("pi_A (set %d)", j + 1); } }
   }
   if (!p->fix_alpha && !(p->seqtype == 1 && p->nssites)) { if (p->malpha) { for (j = 0; j < p->ngene; j++) NAME("alpha (gene %d)", j + 1); } else NAME("alpha"); }
   if (!p->fix_rho) NAME("rho");
#undef NAME
   snprintf(buf, cap, "x%d", i);
   return 0;
}

static void newick_rec(const pamlh *p, int node, char **w, char *end)
{
   int j;
   if (node < p->ns) *w += snprintf(*w, end - *w, "%s", p->names[node]);
   else {
      *w += snprintf(*w, end - *w, "(");
      for (j = p->sons_ptr[node]; j < p->sons_ptr[node + 1]; j++) {
         if (j > p->sons_ptr[node]) *w += snprintf(*w, end - *w, ", ");
         newick_rec(p, p->sons[j], w, end);
      }
      *w += snprintf(*w, end - *w, ")");
   }
   if (node != p->root) {
      if (p->label[node]) *w += snprintf(*w, end - *w, " #%d", p->label[node]);
      *w += snprintf(*w, end - *w, ": %.6f", p->branch[node]);
   }
}

/* The tree in Newick form with the branch lengths of the current model state (pamlh_set_x) and the '#' labels it was read with. */
int pamlh_newick(const pamlh *p, char *buf, int cap)
{
   char *w = buf;
   if (cap < 64 * p->nnode + 128) return -1;
   newick_rec(p, p->root, &w, buf + cap);
   snprintf(w, buf + cap - w, ";");
   return 0;
}

/* The reference's objective-function seam, `double (*com.plfun)(double x[], int np)` (codeml.c:125, baseml.c:70): SetParameters(x)
 * followed by the likelihood evaluation, returning MINUS lnL — the value ming2 minimises — so a driver written against com.plfun
 * can call this instead.  Errors (a parameter vector the model rejects, no GPU) give +1e300 with the message in pamlh_error. */
double pamlh_plfun(pamlh *p, const double *x, int np)
{
   double lnL;
   if (pamlh_set_x(p, x, np) || !pamlh_model_feasible(p) || pamlh_eval_gpu(p, &lnL, NULL)) return 1e300;
   return -lnL;
}

/* Marginal ancestral reconstruction at one node (PostProbNode treesub.c:6142, AncestralMarginal 6288) at the current model state:
 * post[npatt][n] = Pr(state at `node` | pattern).  The engine walks the tree rooted at the node in one fused pass. */
int pamlh_node_posterior(pamlh *p, int node, double *post)
{
   double lnL;
   int rc;
   if (node < p->ns || node >= p->nnode) return pamlh_fail(p, "node %d is not an internal node", node);
   if ((rc = pamlh_eval_gpu(p, &lnL, NULL))) return rc;       /* uploads pi, eigen systems and classes of the current state */
   if ((rc = paml_amd_node_posterior(p->eng, node, p->branch, p->ngene > 1 ? p->rgene : NULL, post))) return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   return 0;
}

/* Joint ancestral reconstruction (AncestralJointPPSG2000 treesub.c; Pupko et al. 2000): for every pattern the assignment of
 * states to ALL internal nodes with the highest probability, and that probability.  Max-product dynamic programming over the
 * tree with the P(t) the device built for the current model state (paml_amd_get_pmat) and the pattern likelihoods of the
 * same evaluation; one rate class only, as in the reference.  O(n_patt x nodes x n^2) on the host — the reconstruction
 * itself is a table walk, not a hot loop.  states[n_patt][n_nodes - n_tips] (internal nodes in node order), prob[n_patt]. */
int pamlh_joint_reconstruction(pamlh *p, int *states, double *prob)
{
   const int n = p->n, ns = p->ns, nn = p->nnode, np = p->npatt, ni = nn - ns;
   double lnL, *lnf, *P, *L, *best_of;
   int *choice, *order, no = 0, h, i, x, y, j, rc;
   if (p->K != 1 || p->ngene > 1) return pamlh_fail(p, "the joint reconstruction needs a model with one rate class and one gene");
   lnf = (double *)malloc(np * sizeof(double));
   if ((rc = pamlh_eval_gpu(p, &lnL, lnf))) { free(lnf); return rc; }
   P = (double *)malloc((size_t)nn * n * n * sizeof(double));
   for (i = 0; i < nn; i++)
      if (i != p->root && (rc = paml_amd_get_pmat(p->eng, 0, 0, i, P + (size_t)i * n * n))) { free(lnf); free(P); return pamlh_fail(p, "%s", paml_amd_last_error(p->eng)); }
   L = (double *)malloc((size_t)nn * n * sizeof(double));        /* L[node][x]: best of the subtree below `node` given state x above it */
   choice = (int *)malloc((size_t)nn * n * sizeof(int));
   best_of = (double *)malloc(n * sizeof(double));
   order = (int *)malloc(nn * sizeof(int));                      /* post-order */
   {
      int *stack = (int *)malloc(2 * nn * sizeof(int)), sp = 0;
      stack[sp++] = p->root;
      while (sp) {                                               /* reverse pre-order = a post-order */
         const int v = stack[--sp];
         order[no++] = v;
         for (j = p->sons_ptr[v]; j < p->sons_ptr[v + 1]; j++) stack[sp++] = p->sons[j];
      }
      free(stack);
   }
   for (h = 0; h < np; h++) {
      for (i = no - 1; i >= 0; i--) {                            /* children before parents */
         const int v = order[i];
         if (v < ns) {                                           /* tip: its state, or the best of its ambiguity set */
            const int code = p->z[(size_t)v * np + h], nc = p->n_chara[code];
            const unsigned char *set = p->chara_map + (size_t)code * n;
            for (x = 0; x < n; x++) {
               double b = -1; int by = 0;
               for (j = 0; j < nc; j++) { const double t = P[((size_t)v * n + x) * n + set[j]]; if (t > b) { b = t; by = set[j]; } }
               L[(size_t)v * n + x] = b; choice[(size_t)v * n + x] = by;
            }
            continue;
         }
         for (y = 0; y < n; y++) {                               /* product over the sons given this node is in state y */
            double t = 1;
            for (j = p->sons_ptr[v]; j < p->sons_ptr[v + 1]; j++) t *= L[(size_t)p->sons[j] * n + y];
            best_of[y] = t;
         }
         if (v == p->root) {
            double b = -1; int by = 0;
            for (y = 0; y < n; y++) { const double t = p->pi[y] * best_of[y]; if (t > b) { b = t; by = y; } }
            prob[h] = b / exp(lnf[h]);
            states[(size_t)h * ni + (v - ns)] = by;
         }
         else
            for (x = 0; x < n; x++) {
               double b = -1; int by = 0;
               for (y = 0; y < n; y++) { const double t = P[((size_t)v * n + x) * n + y] * best_of[y]; if (t > b) { b = t; by = y; } }
               L[(size_t)v * n + x] = b; choice[(size_t)v * n + x] = by;
            }
      }
      for (i = 1; i < no; i++) {                                 /* trace back, parents before children */
         const int v = order[i];
         if (v >= ns) states[(size_t)h * ni + (v - ns)] = choice[(size_t)v * n + states[(size_t)h * ni + (p->father[v] - ns)]];
      }
   }
   free(lnf); free(P); free(L); free(choice); free(best_of); free(order);
   return 0;
}

/* Naive empirical Bayes posteriors of the site classes (lfunNSsites_rate codeml.c:5241-5330) at the current model state
 * (pamlh_set_x): post[k][h] = freqK_k f(x_h | class k) / sum_j freqK_j f(x_h | class j) straight from the device's fhK.
 * post: [K][npatt].  For NSsites models mean_w[h] (may be NULL) gets the posterior mean omega of the pattern. */
int pamlh_neb(pamlh *p, double *post, double *mean_w)
{
   const int K = p->K, np = p->npatt;
   double lnL, *fhK = (double *)malloc((size_t)K * np * sizeof(double));
   int i, h, k, rc, logf = 0;
   if (p->mode != PAML_AMD_MODE_LFUNDG) { free(fhK); return pamlh_fail(p, "NEB needs a model with site classes"); }
   for (i = 0; p->scale && i < p->nnode; i++) if (p->scale[i]) logf = 1;      /* fhK = log f + scale factors (fx_r treesub.c:7744-7749) */
   if ((rc = pamlh_engine_ready(p))) { free(fhK); return rc; }
   if ((rc = paml_amd_set_pi(p->eng, 1, p->pi))) { free(fhK); return pamlh_fail(p, "%s", paml_amd_last_error(p->eng)); }
   if ((rc = pamlh_upload_eigen_sets(p, p->eng, 0, NULL))) { free(fhK); return rc; }
   if ((rc = paml_amd_set_classes(p->eng, p->mode, K, p->freqK, p->rate, p->n_labels, p->eigen_of, p->use_qf ? p->qfactor : NULL)) ||
       (rc = paml_amd_eval(p->eng, p->branch, NULL, &lnL, NULL, fhK))) {
      free(fhK);
      return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   }
   for (h = 0; h < np; h++) {
      double s = 0, mw = 0;
      if (logf) {      /* lfunNSsites_rate codeml.c:5277-5284: relative to the largest class */
         double mx = fhK[h];
         for (k = 1; k < K; k++) if (fhK[(size_t)k * np + h] > mx) mx = fhK[(size_t)k * np + h];
         for (k = 0; k < K; k++) fhK[(size_t)k * np + h] = exp(fhK[(size_t)k * np + h] - mx);
      }
      for (k = 0; k < K; k++) s += p->freqK[k] * fhK[(size_t)k * np + h];
      for (k = 0; k < K; k++) {
         post[(size_t)k * np + h] = s > 0 ? p->freqK[k] * fhK[(size_t)k * np + h] / s : 0;
         mw += post[(size_t)k * np + h] * ((p->seqtype == 1 && p->nssites) ? p->class_w[k] : p->rate[k]);      /* omega classes, or gamma rate classes (lfunRates treesub.c:5850) */
      }
      if (mean_w) mean_w[h] = mw;
   }
   free(fhK);
   return 0;
}

/* Bayes empirical Bayes (Yang, Wong & Nielsen 2005; lfunNSsites_M2M8 codeml.c:6387-6623) for M2a and M8 at the estimates
 * x: the uncertainty of the class parameters is integrated out over a 10^4-point grid with uniform priors —
 *   M2a: (p0, p1) on the 100 triangles of the ternary graph (GetIndexTernary tools.c:4602), w0 ~ U(0,1), w2 ~ U(1,11);
 *   M8:  p0 ~ U(0,1), p, q ~ U(0,2) (class proportions from the beta cdf over ten equal omega bins), ws ~ U(1,11).
 * Stage 1 — f(x_h | w) for the 21 (20) grid omegas at the estimated branch lengths and kappa, scaled by the model's
 * Qfactor_NS — is one evaluation on the device; stage 2, the grid sums, is paml_amd_beb_grid on the same device data.
 * Outputs per pattern: pr_pos (posterior probability of the w > 1 class), mean_w, se_w. */
int pamlh_beb(pamlh *p, const double *x, double *pr_pos, double *mean_w, double *se_w)
{
   enum { N1 = 10 };
   const int np = p->npatt, m2a = p->nssites == 2;
   const int ncls = m2a ? 3 : N1 + 1, K = m2a ? 2 * N1 + 1 : 2 * N1, ngrid = N1 * N1 * N1 * N1;
   double rK[2 * N1 + 1], para[4][N1], lnL, fX, *fhK, *pcl, *lnfXs, *Q, kappa, mr;
   int *iw, i, k, g, rc;
   if (!(p->seqtype == 1 && p->model == 0 && !p->m2a_rel && (p->nssites == 2 || p->nssites == 8))) return pamlh_fail(p, "BEB is defined for NSsites 2 (M2a) and 8 (M8)");
   if ((rc = pamlh_set_x(p, x, p->np))) return rc;
   kappa = p->kappa; mr = p->ns_mr;
   /* the grid (get_grid_para_like_M2M8 codeml.c:6250-6275): bin mid-points */
   for (i = 0; i < N1; i++) {
      para[0][i] = (i + 0.5) / N1;                       /* p0 (M8) */
      para[1][i] = 2 * (i + 0.5) / N1;                   /* p of the beta */
      para[2][i] = m2a ? (i + 0.5) / N1 : 2 * (i + 0.5) / N1;   /* w0 (M2a) or q */
      para[3][i] = 1 + 10 * (i + 0.5) / N1;              /* w2 / ws */
   }
   k = 0;
   for (i = 0; i < N1; i++) rK[k++] = (i + 0.5) / N1;    /* w0 of M2a, or the beta's omega bins */
   if (m2a) rK[k++] = 1;
   for (i = 0; i < N1; i++) rK[k++] = 1 + 10 * (i + 0.5) / N1;
   /* stage 1: f(x_h | w) for every grid omega (fx_r with BayesEB = 1: the model's branch lengths, kappa and Qfactor_NS) */
   Q = (double *)malloc((size_t)p->n * p->n * sizeof(double));
   for (k = 0; k < K; k++) {
      codon_q(p, kappa, rK[k], Q);
      set_eig_uvroot(p, k, Q, p->pi, mr);
      p->freqK[k] = 1.0 / K; p->rate[k] = 1; p->eigen_of[k] = k;
   }
   free(Q);
   p->K = K; p->n_eigen = K; p->mode = PAML_AMD_MODE_LFUNDG;
   fhK = (double *)malloc((size_t)K * np * sizeof(double));
   if ((rc = pamlh_engine_ready(p))) { free(fhK); return rc; }
   rc = paml_amd_set_pi(p->eng, 1, p->pi);
   if (!rc && pamlh_upload_eigen_sets(p, p->eng, 0, NULL)) { free(fhK); return -1; }      /* the K grid omegas: one batch of decompositions on the device */
   if (!rc) rc = paml_amd_set_classes(p->eng, p->mode, K, p->freqK, p->rate, 1, p->eigen_of, NULL);
   if (!rc) rc = paml_amd_eval(p->eng, p->branch, NULL, &lnL, NULL, NULL);      /* fhK stays on the device */
   if (rc) { free(fhK); return pamlh_fail(p, "%s", paml_amd_last_error(p->eng)); }
   /* class proportions and omega index of every class at every grid point (get_pclassM_iw_M2M8 codeml.c:6340-6383) */
   pcl = (double *)malloc((size_t)ngrid * ncls * sizeof(double));
   iw = (int *)malloc((size_t)ngrid * ncls * sizeof(int));
   lnfXs = (double *)malloc(ngrid * sizeof(double));
   for (g = 0; g < ngrid; g++) {
      const int ip0 = g / 1000, ip1 = (g / 100) % 10, ip2 = (g / 10) % 10, ip3 = g % 10;
      if (m2a) {
         const int tri = ip0 * N1 + ip1, ix = (int)sqrt((double)tri), iy = tri - ix * ix;
         const double p0 = (1 + (iy / 2) * 3 + (iy % 2)) / (3.0 * N1), p1 = (1 + (N1 - 1 - ix) * 3 + (iy % 2)) / (3.0 * N1);
         pcl[g * 3] = p0; pcl[g * 3 + 1] = p1; pcl[g * 3 + 2] = 1 - p0 - p1;
         iw[g * 3] = ip2; iw[g * 3 + 1] = N1; iw[g * 3 + 2] = N1 + 1 + ip3;
      }
      else {
         const double p0 = para[0][ip0], bp = para[1][ip1], bq = para[2][ip2];
         for (k = 0; k < N1; k++) {
            const double c0 = k > 0 ? pamlh_betai(bp, bq, k / (double)N1) : 0, c1 = k < N1 - 1 ? pamlh_betai(bp, bq, (k + 1.0) / N1) : 1;
            pcl[g * ncls + k] = p0 * (c1 - c0);
            iw[g * ncls + k] = k;
         }
         pcl[g * ncls + N1] = 1 - p0;
         iw[g * ncls + N1] = N1 + ip3;
      }
   }
   /* stage 2 on the device as well: log f(X | grid point), the marginal likelihood over the grid and the per-pattern
    * posterior sums (codeml.c:6482-6580) — paml_amd_beb_grid over the fhK the evaluation just left on the device */
   if ((rc = paml_amd_beb_grid(p->eng, ngrid, ncls, pcl, iw, rK, &fX, pr_pos, mean_w, se_w))) {
      free(fhK); free(pcl); free(iw); free(lnfXs);
      return pamlh_fail(p, "%s", paml_amd_last_error(p->eng));
   }
   free(fhK); free(pcl); free(iw); free(lnfXs);
   return pamlh_set_x(p, x, p->np);      /* back to the model's own classes */
}

/* Bayes empirical Bayes under branch-site model A and clade models C and D with two branch types (lfunNSsites_ACD
 * codeml.c:6827-7010; Yang, Wong & Nielsen 2005 tables 1 and 2).  Priors: p0,p1 ~ Dir(1,1,1) on the ternary grid; A: w0 ~ U(0,1),
 * w2 ~ U(1,11); C: w0 ~ U(0,1), w2, w3 ~ U(0,3); D: w0 ~ U(0,1), w1 ~ U(0.01,1.5), w2, w3 ~ U(0,3); ten bins each, so the grid has
 * 10^4 (A), 10^5 (C) or 10^6 (D) points.  Branch lengths, kappa and the branch-type time scales stay at the estimates x.
 * f(x_h | omega of type 0, omega of type 1) is needed for 121 (A: 10 w0, w1 = 1, 10 x 10 (w0, w2), 10 (1, w2)), 111 (C: 10 w0, 1,
 * 10 x 10 (w2, w3)) or 120 (D: 10 w0, 10 w1, 10 x 10) pairs: ONE evaluation with that many classes over 21 / 30 eigen systems, on
 * an engine of its own (the analysis' engine is sized for the model's classes).
 * post[nc][npatt], nc = 4 (A: classes 0, 1, 2a, 2b) or 3 (C, D). */
int pamlh_beb_acd(pamlh *p, const double *x, double *post)
{
   enum { N1 = 10, KWMAX = 3 * N1, KCMAX = 2 * N1 + N1 * N1 + N1 };
   const int n = p->n, mA = p->model == 2, mD = p->model == 3 && p->nssites == 3;
   const int nc = mA ? 4 : 3, dim = mA ? 4 : mD ? 6 : 5;
   const int KW = mA ? 2 * N1 + 1 : mD ? 3 * N1 : 2 * N1 + 1, KC = mA ? N1 + 1 + N1 * N1 + N1 : mD ? 2 * N1 + N1 * N1 : N1 + 1 + N1 * N1;
   paml_amd_engine *e = NULL;
   double wv[KWMAX], qf[KCMAX * 2], fk[KCMAX], rt[KCMAX], lnL, fX, *Q, *U, *V, *R, *pcl = NULL, kappa;
   int eo[KCMAX * 2], *iw = NULL, i, k, rc;
   long g, ng = 1;
   if (!(p->seqtype == 1 && ((p->model == 2 && p->nssites == 2) || (p->model == 3 && p->n_omega == 2))))
      return pamlh_fail(p, "this BEB is defined for branch-site model A and for clade models C and D with two branch types");
   if ((rc = pamlh_set_x(p, x, p->np))) return rc;
   kappa = p->kappa;
   for (i = 0; i < dim; i++) ng *= N1;
   /* the omega values on the grid, and the pair of them each evaluated class uses (get_grid_para_like_ACD codeml.c:6629-6720) */
   for (i = 0; i < N1; i++) wv[i] = (i + 0.5) / N1;                                               /* w0 */
   if (mD) for (i = 0; i < N1; i++) wv[N1 + i] = 0.01 + (i + 0.5) * 1.49 / N1; else wv[N1] = 1;     /* w1 */
   for (i = 0; i < N1; i++) wv[KW - N1 + i] = mA ? 1 + 10 * (i + 0.5) / N1 : 3 * (i + 0.5) / N1;   /* w2 (A) / w2, w3 (C, D) */
   for (k = 0; k < KC; k++) {
      const int n01 = mD ? 2 * N1 : N1 + 1;      /* classes of the first two site classes */
      int b, f;
      if (k < n01) b = f = k;                    /* site classes 0 and 1: the same omega on every branch */
      else if (mA && k < n01 + N1 * N1) { b = (k - n01) / N1; f = KW - N1 + (k - n01) % N1; }   /* 2a: w0 back, w2 fore */
      else if (mA) { b = N1; f = KW - N1 + (k - n01 - N1 * N1); }                              /* 2b: 1 back, w2 fore */
      else { b = KW - N1 + (k - n01) / N1; f = KW - N1 + (k - n01) % N1; }                       /* clade: type 0, type 1 */
      eo[k * 2] = b; eo[k * 2 + 1] = f;
      qf[k * 2] = p->qfactor[0]; qf[k * 2 + 1] = p->qfactor[1];      /* Qfactor_NS_branch stays at the MLE (treesub.c:7556-7566) */
      fk[k] = 1.0 / KC; rt[k] = 1;
   }
   if ((rc = paml_amd_create(&e, n, p->ns, p->npatt, KC, 1, 0))) return pamlh_fail(p, "paml_amd_create failed (%d)", rc);
   Q = (double *)malloc((size_t)4 * n * n * sizeof(double)); U = Q + (size_t)n * n; V = U + (size_t)n * n; R = V + (size_t)n * n;
   rc = paml_amd_set_tips(e, p->z, p->cleandata, p->n_codes, p->n_chara, p->chara_map, p->w, NULL);
   if (!rc) rc = paml_amd_set_tree(e, p->nnode, p->root, p->sons_ptr, p->sons, p->label, p->scale);
   if (!rc) rc = paml_amd_set_pi(e, 1, p->pi);
   if (host_eigen_forced())
      for (k = 0; k < KW && !rc; k++) {
         codon_q(p, kappa, wv[k], Q);
         pamlh_eigen_qrev(Q, p->pi, n, R, U, V);
         rc = paml_amd_set_eigen_uvroot(e, k, U, V, R);
      }
   else if (!rc) {      /* the grid's omegas: one batch of decompositions on the device */
      double *Qs = (double *)malloc((size_t)KW * n * n * sizeof(double)), *pis = (double *)malloc((size_t)KW * n * sizeof(double));
      int ids[KWMAX];
      for (k = 0; k < KW; k++) {
         codon_q(p, kappa, wv[k], Qs + (size_t)k * n * n);
         memcpy(pis + (size_t)k * n, p->pi, n * sizeof(double));
         ids[k] = k;
      }
      rc = paml_amd_set_eigen_qrev_batch(e, KW, ids, Qs, pis, NULL);
      free(Qs); free(pis);
   }
   if (!rc) rc = paml_amd_set_classes(e, PAML_AMD_MODE_LFUNDG, KC, fk, rt, 2, eo, qf);
   if (!rc) rc = paml_amd_eval(e, p->branch, NULL, &lnL, NULL, NULL);
   free(Q);
   if (rc) { pamlh_fail(p, "%s", paml_amd_last_error(e)); paml_amd_destroy(e); return rc; }
   /* proportions and class index of the site classes at every grid point (get_pclassM_iw_ACD codeml.c:6760-6820) */
   pcl = (double *)malloc((size_t)ng * nc * sizeof(double));
   iw = (int *)malloc((size_t)ng * nc * sizeof(int));
   for (g = 0; g < ng; g++) {
      int ip[6], j;
      long it = g;
      for (j = dim - 1; j >= 0; j--) { ip[j] = (int)(it % N1); it /= N1; }
      {
         const int tri = ip[0] * N1 + ip[1], ix = (int)sqrt((double)tri), iy = tri - ix * ix;
         const double p0 = (1 + (iy / 2) * 3 + (iy % 2)) / (3.0 * N1), p1 = (1 + (N1 - 1 - ix) * 3 + (iy % 2)) / (3.0 * N1), p2 = 1 - p0 - p1;
         pcl[g * nc] = p0; pcl[g * nc + 1] = p1;
         if (mA) { pcl[g * nc + 2] = p2 * p0 / (1 - p2); pcl[g * nc + 3] = p2 * p1 / (1 - p2); }
         else pcl[g * nc + 2] = p2;
      }
      iw[g * nc] = ip[2];
      if (mA) { iw[g * nc + 1] = N1; iw[g * nc + 2] = N1 + 1 + ip[2] * N1 + ip[3]; iw[g * nc + 3] = N1 + 1 + N1 * N1 + ip[3]; }
      else if (mD) { iw[g * nc + 1] = N1 + ip[3]; iw[g * nc + 2] = 2 * N1 + ip[4] * N1 + ip[5]; }
      else { iw[g * nc + 1] = N1; iw[g * nc + 2] = N1 + 1 + ip[3] * N1 + ip[4]; }
   }
   rc = paml_amd_beb_grid_classes(e, (int)ng, nc, pcl, iw, &fX, post);
   if (rc) pamlh_fail(p, "%s", paml_amd_last_error(e));
   free(pcl); free(iw);
   paml_amd_destroy(e);
   return rc;
}

/* dN and dS of every branch under the codon models without site classes — M0, the branch models, the free-ratio model — as the
 * reference's "dN & dS for each branch" table (DetailOutput codeml.c:1349-1404; eigenQcodon mode 2 codeml.c:3318-3365): with rs / ra
 * the synonymous / nonsynonymous shares of the mean rate under the branch's omega and rs0 / ra0 the shares at omega = 1,
 *    S = 3 ls rs0,  N = 3 ls - S,  dS = t rs / (3 rs0),  dN = t ra / (3 ra0).
 * out: [n_branches][6] = t, N, S, omega, dN, dS in the order of the branch lengths in x. */
int pamlh_dnds(pamlh *p, const double *x, double *out)
{
   const int n = p->n;
   double *Q, *Q1;
   int from61[64], b, i, j, m = 0, rc;
   if (p->seqtype != 1 || p->nssites || p->aadist || p->ngene > 1 || !(p->model == 0 || p->model == 2))
      return pamlh_fail(p, "dN and dS per branch are defined for the codon models without site classes (one gene)");
   if ((rc = pamlh_set_x(p, x, p->np))) return rc;
   for (i = 0; i < 64; i++) if (p->code[i] != '*') from61[m++] = i;
   Q = (double *)malloc((size_t)2 * n * n * sizeof(double)); Q1 = Q + (size_t)n * n;
   codon_q(p, p->kappa, 1, Q1);
   for (b = 0; b < p->nbranch; b++) {
      const int node = p->branch_node[b];
      const double w = p->model == 2 ? p->class_w[p->label[node]] : p->omega, t = p->branch[node];
      double rs = 0, ra = 0, ra0 = 0, mr, rs0;
      codon_q(p, p->kappa, w, Q);
      for (i = 0; i < n; i++)
         for (j = 0; j < n; j++) {
            if (i == j) continue;
            if (p->code[from61[i]] == p->code[from61[j]]) rs += p->pi[i] * Q[i * n + j];
            else { ra += p->pi[i] * Q[i * n + j]; ra0 += p->pi[i] * Q1[i * n + j]; }
         }
      mr = rs + ra;
      rs0 = rs / (rs + ra0); ra0 = ra0 / (rs + ra0);
      out[b * 6 + 0] = t;
      out[b * 6 + 2] = 3 * p->ls * rs0;
      out[b * 6 + 1] = 3 * p->ls - out[b * 6 + 2];
      out[b * 6 + 3] = w;
      out[b * 6 + 4] = t * (ra / mr) / (3 * ra0);
      out[b * 6 + 5] = t * (rs / mr) / (3 * rs0);
   }
   free(Q);
   return 0;
}

const int *pamlh_pose(const pamlh *p, int *n_sites)
{
   if (n_sites) *n_sites = p->n_pose;
   return p->pose;
}

const double *pamlh_class_omega(const pamlh *p) { return p->class_w; }
const double *pamlh_qfactor(const pamlh *p) { return p->use_qf ? p->qfactor : NULL; }
/* auto-discrete-gamma: the K x K rate-class transition matrix of the current model state (NULL when the model has none) */
const double *pamlh_adg_matrix(const pamlh *p) { return p->adg ? p->MK : NULL; }

/* option G: number of genes, first pattern of each (n_genes + 1 entries), their rates (after pamlh_set_x), the number of
 * frequency vectors pamlh_pi holds (1 or n_genes) and the eigen system of (gene, class) */
int pamlh_genes(const pamlh *p, const int **gene_off, const double **gene_rate, int *n_pi, const int **gene_eigen_of)
{
   if (gene_off) *gene_off = p->posG;
   if (gene_rate) *gene_rate = p->rgene;
   if (n_pi) *n_pi = p->n_pi;
   if (gene_eigen_of) *gene_eigen_of = p->ngene > 1 ? p->gene_eigen_of : NULL;
   return p->ngene;
}

/* how many of the last site classes allow omega > 1 (the classes whose posterior the reference's NEB table sums):
 * 1 for M2a, M8 and the clade models, 2 for the branch-site models (classes 2a + 2b), 0 for models without such a class */
int pamlh_positive_classes(const pamlh *p)
{
   if (p->seqtype != 1 || !p->nssites || p->mode != PAML_AMD_MODE_LFUNDG) return 0;
   if (p->model == 2) return 2;
   if (p->model == 3 || p->nssites == 2 || p->nssites == 8 || p->nssites == 3 || p->nssites == 4) return 1;
   return 0;
}

int pamlh_write_lnf(const pamlh *p, const char *path, const double *lnf)
{
   FILE *f = fopen(path, "w");
   int h, j;
   if (!f) return -1;
   fprintf(f, "%6d %6d %6d\n\n\n%2d\n\n", 1, p->ls, p->npatt, 1);
   for (h = 0; h < p->npatt; h++) {
      fprintf(f, "%6d %6.0f %16.10f %16.12f %12.4f  ", h + 1, p->w[h], lnf[h], exp(lnf[h]), p->ls * exp(lnf[h]));
      for (j = 0; j < p->ns; j++) {
         fwrite(p->raw + ((size_t)j * p->npatt + h) * p->n31, 1, p->n31, f);
         if (p->n31 == 3) fputc(' ', f);
      }
      fputc('\n', f);
   }
   fclose(f);
   return 0;
}
