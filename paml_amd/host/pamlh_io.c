/* pamlh_io.c — reads what the reference programs read: control file, sequence file, tree file.
 * Written fresh; behaviour follows GetOptions (codeml.c:1694-1886, baseml.c:954-1146: "key = value" lines,
 * '*' and '#' start comments, keys compared on their first 8 characters), ReadSeq (treesub.c:487-999),
 * RemoveIndel (1754), PatternWeight (1386-1516: patterns in sorted order of the raw characters),
 * EncodeSeqs (1116-1187), SetMapAmbiguity (1218-1286), ReadTreeN (3048-3216) and SetNodeScale (7177-7197). */
#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pamlh_internal.h"

static const char BASEs[] = "TCAGUYRMKSWHBVD-N?";
static const char *EquateBASE[] = {"T", "C", "A", "G", "T", "TC", "AG", "CA", "TG", "CG", "TA", "TCA", "TCG", "CAG", "TAG",
                                   "TCAG", "TCAG", "TCAG"};
static const char AAs[] = "ARNDCQEGHILKMFPSTWYV-*?X";
/* standard genetic code, codon index 16 b1 + 4 b2 + b3 with T,C,A,G = 0..3 (tools.c:23-84) */

int pamlh_fail(pamlh *p, const char *fmt, ...)
{
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(p->err, sizeof(p->err), fmt, ap);
   va_end(ap);
   return -1;
}

/* ------------------------------------------------------------------ control file */
int pamlh_read_ctl(pamlh *p, const char *path)
{
   FILE *f = fopen(path, "r");
   char line[4096];
   if (!f) return pamlh_fail(p, "cannot open control file %s", path);
   p->ctl.n = 0;
   while (fgets(line, sizeof(line), f)) {
      char *c = line, *eq, *k, *v, *e;
      for (; *c; c++)
         if (*c == '*') { *c = 0; break; }      /* '*' starts a comment (the reference's GetOptions, codeml.c:1722); '#' does not */
      eq = strchr(line, '=');
      if (!eq) continue;
      *eq = 0;
      k = line;
      while (isspace((unsigned char)*k)) k++;
      e = k + strlen(k);
      while (e > k && isspace((unsigned char)e[-1])) *--e = 0;
      v = eq + 1;
      while (isspace((unsigned char)*v)) v++;
      e = v + strlen(v);
      while (e > v && isspace((unsigned char)e[-1])) *--e = 0;
      if (!*k) continue;
      {      /* an option given twice: the later line wins, as in the reference's sequential GetOptions */
         int i;
         for (i = 0; i < p->ctl.n; i++) if (strncmp(p->ctl.key[i], k, 8) == 0) break;
         if (i == p->ctl.n) { if (p->ctl.n >= PAMLH_MAXOPT) continue; p->ctl.n++; }
         snprintf(p->ctl.key[i], 32, "%s", k);
         snprintf(p->ctl.val[i], 1024, "%s", v);
         /* file-name options: the reference reads ONE token (sscanf "%s", codeml.c:1735) — anything after it on the line is ignored */
         if (!strncmp(k, "seqfile", 7) || !strncmp(k, "treefile", 8) || !strncmp(k, "outfile", 7) || !strncmp(k, "aaRatefile", 10)) {
            char *t = p->ctl.val[i];
            while (*t && !isspace((unsigned char)*t)) t++;
            *t = 0;
         }
      }
   }
   fclose(f);
   return 0;
}

/* "key = value" lines (newline or ';' between them) replace the control file's options of the same name, or are added */
int pamlh_ctl_override(pamlh *p, const char *text)
{
   const char *c = text;
   while (*c) {
      char line[1200], *eq, *k, *v, *e;
      size_t n = strcspn(c, "\n;");
      int i;
      if (n >= sizeof(line)) return pamlh_fail(p, "option override too long");
      memcpy(line, c, n); line[n] = 0;
      c += n + (c[n] != 0);
      eq = strchr(line, '=');
      if (!eq) { for (k = line; isspace((unsigned char)*k); k++) {} if (*k) return pamlh_fail(p, "option override without '=': %s", line); continue; }
      *eq = 0;
      for (k = line; isspace((unsigned char)*k); k++) {}
      for (e = k + strlen(k); e > k && isspace((unsigned char)e[-1]);) *--e = 0;
      for (v = eq + 1; isspace((unsigned char)*v); v++) {}
      for (e = v + strlen(v); e > v && isspace((unsigned char)e[-1]);) *--e = 0;
      if (!*k) return pamlh_fail(p, "option override without a name");
      for (i = 0; i < p->ctl.n; i++) if (strncmp(p->ctl.key[i], k, 8) == 0) break;
      if (i == p->ctl.n) { if (p->ctl.n >= PAMLH_MAXOPT) return pamlh_fail(p, "too many options"); p->ctl.n++; }
      snprintf(p->ctl.key[i], 32, "%s", k);
      snprintf(p->ctl.val[i], 1024, "%s", v);
   }
   return 0;
}

const char *pamlh_opt(const pamlh *p, const char *key)
{
   int i;
   for (i = 0; i < p->ctl.n; i++)
      if (strncmp(p->ctl.key[i], key, 8) == 0) return p->ctl.val[i];   /* first 8 characters decide (codeml.c:1730) */
   return NULL;
}

double pamlh_optd(const pamlh *p, const char *key, double dflt)
{
   const char *v = pamlh_opt(p, key);
   return v && *v ? atof(v) : dflt;
}

/* ------------------------------------------------------------------ sequences */
static const int *g_sort_gene;      /* gene of every (kept) site, NULL with one gene */
static int cmp_cols(const void *a, const void *b, void *ctx)
{
   const pamlh *p = (const pamlh *)ctx;
   const int ia = *(const int *)a, ib = *(const int *)b, w = p->n31, L = p->ls * w;
   int j, c;
   if (g_sort_gene && g_sort_gene[ia] != g_sort_gene[ib]) return g_sort_gene[ia] - g_sort_gene[ib];   /* patterns are sorted within genes */
   for (j = 0; j < p->ns; j++) {
      c = memcmp(p->raw + (size_t)j * L + (size_t)ia * w, p->raw + (size_t)j * L + (size_t)ib * w, w);
      if (c) return c;
   }
   return ia - ib;
}
static const pamlh *g_sort_ctx;
static int cmp_cols0(const void *a, const void *b) { return cmp_cols(a, b, (void *)g_sort_ctx); }

static int base_set(char c, int *set)
{
   const char *q = strchr(BASEs, c);
   int i, n;
   if (!q) return 0;
   n = (int)strlen(EquateBASE[q - BASEs]);
   for (i = 0; i < n; i++) set[i] = (int)(strchr(BASEs, EquateBASE[q - BASEs][i]) - BASEs);
   return n;
}

/* The sequence file as a stream in the native format.  Aligned FASTA ('>' first) and NEXUS ("begin data" ... ntax= nchar= ...
 * "matrix") files (GetSeqFileType / ScanFastaFile treesub.c:367-419) are rewritten in memory as "ns ls" + one
 * "name  sequence" line per sequence, which the reader below takes as sequential PHYLIP. */
static FILE *open_seqfile(pamlh *p, char **mem)
{
   FILE *f = fopen(p->seqfile, "r");
   long len;
   char *txt, *out, *q, *w;
   int ns = 0, ls = -1, a, b;
   *mem = NULL;
   if (!f) return NULL;
   fseek(f, 0, SEEK_END); len = ftell(f); rewind(f);
   txt = (char *)malloc(len + 1);
   if (fread(txt, 1, len, f) != (size_t)len) { free(txt); rewind(f); return f; }
   txt[len] = 0;
   for (q = txt; *q && isspace((unsigned char)*q); q++) ;
   if (*q != '>' && sscanf(q, "%d%d", &a, &b) == 2) { free(txt); rewind(f); return f; }      /* native format */
   fclose(f);
   out = (char *)malloc(2 * len + 64);
   w = out + 32;                        /* room for the "ns ls" line in front */
   if (*q == '>') {                     /* FASTA: >name, then the sequence over any number of lines */
      while (*q == '>') {
         int n = 0;
         for (q++; *q && *q != '\n' && *q != '\r'; q++) *w++ = isspace((unsigned char)*q) ? '_' : *q;
         *w++ = ' '; *w++ = ' ';
         for (; *q && *q != '>'; q++)
            if (!isspace((unsigned char)*q)) { *w++ = *q; n++; }
         *w++ = '\n';
         if (ls >= 0 && n != ls) { pamlh_fail(p, "the seq file appears to be in fasta format, but not aligned (sequence %d has %d characters, the first %d)", ns + 1, n, ls); free(txt); free(out); return NULL; }
         ls = n; ns++;
      }
   }
   else {                               /* NEXUS */
      char *low = (char *)malloc(len + 1), *m, *e;
      for (a = 0; a <= len; a++) low[a] = (char)tolower((unsigned char)txt[a]);
      m = strstr(low, "begin data");
      if (m && (e = strstr(m, "ntax"))) { e = strchr(e, '='); if (e) ns = atoi(e + 1); }
      if (m && (e = strstr(m, "nchar"))) { e = strchr(e, '='); if (e) ls = atoi(e + 1); }
      m = m ? strstr(m, "matrix") : NULL;
      if (!m || ns < 1 || ls < 1) { pamlh_fail(p, "%s is neither PHYLIP, FASTA nor NEXUS (begin data / ntax= nchar= / matrix)", p->seqfile); free(txt); free(out); free(low); return NULL; }
      q = txt + (m - low) + 6;
      while (*q && *q != '\n') q++;     /* the rest of the "matrix" line */
      for (a = 0; a < ns; a++) {        /* name, then characters until ls are in; [comments] skipped */
         int n = 0;
         for (;;) {                     /* white space and [comments] in front of the name */
            while (*q && isspace((unsigned char)*q)) q++;
            if (*q != '[') break;
            while (*q && *q != ']') q++;
            if (*q) q++;
         }
         while (*q && !isspace((unsigned char)*q)) *w++ = *q++;
         *w++ = ' '; *w++ = ' ';
         for (; *q && n < ls; q++) {
            if (*q == '[') { while (*q && *q != ']') q++; continue; }
            if (!isspace((unsigned char)*q)) { *w++ = *q; n++; }
         }
         *w++ = '\n';
         if (n != ls) { pamlh_fail(p, "NEXUS matrix: sequence %d has %d of %d characters", a + 1, n, ls); free(txt); free(out); free(low); return NULL; }
      }
      free(low);
   }
   *w = 0;
   {
      char head[32];
      const int hl = snprintf(head, sizeof(head), "%d %d\n", ns, ls);
      memcpy(out + 32 - hl, head, hl);
      *mem = out;
      free(txt);
      return fmemopen(out + 32 - hl, (size_t)(w - (out + 32 - hl)), "r");
   }
}

int pamlh_read_seqs(pamlh *p)
{
   char *seqmem = NULL;
   FILE *f = open_seqfile(p, &seqmem);
   char *line;
   size_t cap = 1 << 16;
   int ns, lsraw, i, j, k, h, readpattern = 0, interleaved = 0, any_amb = 0, n31 = (p->seqtype == 1 || p->translate ? 3 : 1);
   int *pos, *site_gene = NULL;
   const char *alpha = (p->seqtype == 2 && !p->translate) ? AAs : BASEs;
   int nbasic = (p->seqtype == 2 && !p->translate) ? 20 : 4;
   char *seq;
   if (!f) return p->err[0] ? -1 : pamlh_fail(p, "cannot open sequence file %s", p->seqfile);
   line = (char *)malloc(cap);
   if (!fgets(line, (int)cap, f) || sscanf(line, "%d %d", &ns, &lsraw) != 2) { fclose(f); return pamlh_fail(p, "bad first line in %s", p->seqfile); }
   {  /* option letters after the two numbers */
      char *c = line;
      int nnum = 0, hasG = 0, hasC = 0;
      while (*c) {
         if (isdigit((unsigned char)*c)) { nnum++; while (isdigit((unsigned char)*c)) c++; continue; }
         if (nnum >= 2 && isalpha((unsigned char)*c)) {
            char o = (char)toupper(*c);
            if (o == 'P') readpattern = 1;
            else if (o == 'G') hasG = 1;
            else if (o == 'C') hasC = 1;
            else if (o == 'I') interleaved = 1;
            else if (o != 'S') { fclose(f); free(line); return pamlh_fail(p, "bad option '%c' in first line of seqfile", o); }
         }
         c++;
      }
      if (lsraw % n31) { fclose(f); free(line); return pamlh_fail(p, "%d nucleotides, not a multiple of 3", lsraw); }
      /* option G: sites (codons for codon data) belong to several genes / partitions (ReadSeq treesub.c:590-680).  "GC" is
       * baseml's shorthand for the three codon positions; codeml ignores it.  Otherwise one option line follows:
       * "G ngene" + the gene lengths on the same line, or "G ngene" + one gene mark (1..ngene) per site on the next lines. */
      p->ngene = 1;
      if (hasG && hasC) {
         if (p->seqtype == 0) {
            if (lsraw % 3) { fclose(f); free(line); return pamlh_fail(p, "option GC: %d sites, not a multiple of 3", lsraw); }
            if (readpattern) { fclose(f); free(line); return pamlh_fail(p, "patterns for coding sequences (G C P) are not supported"); }
            p->ngene = 3;
            site_gene = (int *)malloc(lsraw * sizeof(int));
            for (h = 0; h < lsraw; h++) site_gene[h] = h % 3;
         }
      }
      else if (hasG) {
         const int nsite = lsraw / n31;
         int ch, ng = 0, tot = 0;
         char *q;
         /* (with the P format the numbers on the G line are PATTERNS per gene, the patterns of a gene standing together: treesub.c:640-665) */
         do ch = fgetc(f); while (ch != EOF && !isalnum(ch));
         if (toupper(ch) != 'G' || fscanf(f, "%d", &ng) != 1 || ng < 1 || ng > PAMLH_MAXGENE) { fclose(f); free(line); return pamlh_fail(p, "option G: expecting 'G <number of genes (<= %d)>'", PAMLH_MAXGENE); }
         site_gene = (int *)malloc(nsite * sizeof(int));
         if (!fgets(line, (int)cap, f)) line[0] = 0;
         for (q = line; *q && isspace((unsigned char)*q); q++) ;
         if (*q) {                               /* gene lengths */
            for (i = 0; i < ng; i++) {
               int len;
               while (*q && !isdigit((unsigned char)*q)) q++;
               if (*q) { len = atoi(q); while (isdigit((unsigned char)*q)) q++; }
               else if (fscanf(f, "%d", &len) != 1) { fclose(f); free(line); free(site_gene); return pamlh_fail(p, "option G: EOF reading the gene lengths"); }
               if (len < 1 || tot + len > nsite) { fclose(f); free(line); free(site_gene); return pamlh_fail(p, "option G: total length over genes is not correct"); }
               for (h = 0; h < len; h++) site_gene[tot + h] = i;
               tot += len;
            }
            if (tot != nsite) { fclose(f); free(line); free(site_gene); return pamlh_fail(p, "option G: gene lengths sum to %d, not %d%s", tot, nsite, n31 == 3 ? " (gene lengths are in codons)" : ""); }
         }
         else {                                  /* one mark per site */
            if (readpattern) { fclose(f); free(line); free(site_gene); return pamlh_fail(p, "option PG: use the number of patterns in each gene, not site marks"); }
            for (h = 0; h < nsite; h++) {
               int m;
               if (ng > 9) { if (fscanf(f, "%d", &m) != 1) m = -1; }
               else { do ch = fgetc(f); while (ch != EOF && !isdigit(ch)); m = ch == EOF ? -1 : ch - '0'; }
               if (m < 1 || m > ng) { fclose(f); free(line); free(site_gene); return pamlh_fail(p, "option G: gene mark %d at site %d?", m, h + 1); }
               site_gene[h] = m - 1;
            }
            if (!fgets(line, (int)cap, f)) line[0] = 0;      /* the rest of the last line of marks */
         }
         p->ngene = ng;
      }
   }
   p->ns = ns; p->n31 = n31;
   p->names = (char **)calloc(ns, sizeof(char *));
   seq = (char *)malloc((size_t)ns * lsraw);
   pos = (int *)calloc(ns, sizeof(int));
   /* sequential: each sequence runs on over as many lines as it needs.  Interleaved (option I, ReadSeq treesub.c:487):
    * the first block has name + first stretch of every sequence, one line each; the following blocks continue the
    * sequences in the same order without names */
   for (i = 0; i == 0 || (interleaved && pos[ns - 1] < lsraw); i++)
   for (j = 0; j < ns; j++) {
      char *q, *dbl;
      do {
         if (!fgets(line, (int)cap, f)) { fclose(f); return pamlh_fail(p, "EOF reading sequence %d", j + 1); }
         for (q = line; *q && isspace((unsigned char)*q); q++) ;
      } while (!*q);
      dbl = strstr(q, "  ");
      if (i == 0) {
         size_t ln = dbl ? (size_t)(dbl - q) : strcspn(q, "\r\n");
         size_t tl = strcspn(q, "\t\r\n");
         if (tl < ln) ln = tl;
         if (ln > 95) ln = 95;
         p->names[j] = (char *)calloc(ln + 1, 1);
         memcpy(p->names[j], q, ln);
         while (ln > 0 && isspace((unsigned char)p->names[j][ln - 1])) p->names[j][--ln] = 0;
         q += (dbl ? (size_t)(dbl - q) : strlen(q));
      }
      for (k = pos[j]; k < lsraw;) {
         if (!*q) {
            if (interleaved) break;            /* the rest of this sequence is in the next block */
            if (!fgets(line, (int)cap, f)) { fclose(f); return pamlh_fail(p, "EOF at site %d, seq %d", k + 1, j + 1); }
            q = line;
            continue;
         }
         {
            char ch = (char)toupper((unsigned char)*q++);
            if (alpha == BASEs && ch == 'U') ch = 'T';
            if (ch == '.') {
               if (j == 0) { fclose(f); return pamlh_fail(p, ". in the first sequence"); }
               seq[(size_t)j * lsraw + k] = seq[k];
               k++;
            }
            else if (strchr(alpha, ch) && ch) {
               seq[(size_t)j * lsraw + k++] = ch;
               if (strchr(alpha, ch) - alpha >= nbasic) any_amb = 1;
            }
            else if (isalpha((unsigned char)ch)) { fclose(f); return pamlh_fail(p, "bad character %c at %d seq %d", ch, k + 1, j + 1); }
         }
      }
      pos[j] = k;
   }
   for (j = 0; j < ns; j++)
      if (pos[j] != lsraw) { fclose(f); return pamlh_fail(p, "sequence %d has %d of %d characters", j + 1, pos[j], lsraw); }
   free(pos);
   if (p->translate) {
      /* seqtype = 3: every codon becomes its amino acid (DNA2protein / Codon2AA tools.c:777-828): the amino acids of all the codons an
       * ambiguous triplet stands for, stop codons aside; exactly one -> that amino acid, none or several -> '-' (missing) */
      static const char NUCS[] = "TCAGYRMKSWHBVD-N?";
      static const int MASK[] = {1, 2, 4, 8, 3, 12, 6, 9, 10, 5, 7, 11, 14, 13, 15, 15, 15};
      const char *code = pamlh_genetic_code(p->icode);
      const int lc = lsraw / 3;
      if (!code) { fclose(f); return pamlh_fail(p, "genetic code icode = %d is not supported", p->icode); }
      if (site_gene) { fclose(f); return pamlh_fail(p, "seqtype = 3 with option G is not supported"); }
      any_amb = 0;
      for (j = 0; j < ns; j++)
         for (h = 0; h < lc; h++) {
            int m[3], b0, b1, b2, aa = -1, many = 0;
            for (k = 0; k < 3; k++) { const char *q = strchr(NUCS, seq[(size_t)j * lsraw + h * 3 + k]); m[k] = q && *q ? MASK[q - NUCS] : 15; }
            for (b0 = 0; b0 < 4; b0++) for (b1 = 0; b1 < 4; b1++) for (b2 = 0; b2 < 4; b2++)
               if ((m[0] >> b0 & 1) && (m[1] >> b1 & 1) && (m[2] >> b2 & 1)) {
                  const char a = code[b0 * 16 + b1 * 4 + b2];
                  if (a == '*') continue;
                  if (aa < 0) aa = a; else if (a != aa) many = 1;
               }
            seq[(size_t)j * lc + h] = (aa < 0 || many) ? '-' : (char)aa;      /* (compacting in place: position j * lc + h <= j * lsraw + 3 h) */
            if (aa < 0 || many) any_amb = 1;
         }
      lsraw = lc; n31 = 1; alpha = AAs; nbasic = 20;
   }
   /* pattern counts of the P format: npatt numbers after the sequences (treesub.c:954-983) */
   {
      const int nsite = lsraw / n31;
      int *keep = (int *)malloc(nsite * sizeof(int)), nkeep = 0, *idx;
      double *cnt = (double *)malloc(nsite * sizeof(double));
      for (h = 0; h < nsite; h++) cnt[h] = 1;
      if (readpattern)
         for (h = 0; h < nsite; h++)
            if (fscanf(f, "%lf", &cnt[h]) != 1) { fclose(f); return pamlh_fail(p, "EOF reading pattern counts"); }
      fclose(f);
      /* cleandata = 1: drop every site with an ambiguity character in any sequence (RemoveIndel treesub.c:1754) */
      p->cleandata = (p->cleandata_opt || !any_amb) ? 1 : 0;
      for (h = 0; h < nsite; h++) {
         int ok = 1;
         if (p->cleandata_opt && any_amb)
            for (j = 0; j < ns && ok; j++)
               for (k = 0; k < n31; k++)
                  if (strchr(alpha, seq[(size_t)j * lsraw + h * n31 + k]) - alpha >= nbasic) { ok = 0; break; }
         if (ok) keep[nkeep++] = h;
      }
      p->ls = nkeep;
      p->raw = (char *)malloc((size_t)ns * nkeep * n31);
      for (j = 0; j < ns; j++)
         for (h = 0; h < nkeep; h++) memcpy(p->raw + ((size_t)j * nkeep + h) * n31, seq + (size_t)j * lsraw + keep[h] * n31, n31);
      /* compress sites into patterns, sorted by raw characters; the P format is already compressed */
      idx = (int *)malloc(nkeep * sizeof(int));
      for (h = 0; h < nkeep; h++) idx[h] = h;
      {
         char *raw2;
         int np = 0, *first = (int *)malloc(nkeep * sizeof(int));
         double *w = (double *)calloc(nkeep, sizeof(double));
         p->pose = (int *)malloc((nkeep + 1) * sizeof(int));      /* com.pose: site (after cleaning) -> pattern */
         int *kg = NULL;
         if (site_gene) {
            kg = (int *)malloc(nkeep * sizeof(int));
            for (h = 0; h < nkeep; h++) kg[h] = site_gene[keep[h]];
         }
         if (!readpattern && getenv("PAMLH_GPU_COMPRESS") && getenv("PAMLH_GPU_COMPRESS")[0] == '1') {
            /* the same result from the device (paml_amd_compress_patterns: radix sort of the columns); asked for explicitly,
             * so a failure is an error, not a quiet return to the host sort */
            const int rc = paml_amd_compress_patterns(ns, nkeep, n31, (const unsigned char *)p->raw, kg, &np, first, w, p->pose);
            if (rc) return pamlh_fail(p, "PAMLH_GPU_COMPRESS=1 but paml_amd_compress_patterns failed (%d): no GPU?", rc);
         }
         else if (!readpattern) {
            g_sort_ctx = p;
            g_sort_gene = kg;
            qsort(idx, nkeep, sizeof(int), cmp_cols0);
            g_sort_gene = NULL;
            for (h = 0; h < nkeep; h++) {
               int same = 0;
               if (np > 0 && !(kg && kg[idx[h]] != kg[first[np - 1]])) {
                  same = 1;
                  for (j = 0; j < ns && same; j++)
                     if (memcmp(p->raw + ((size_t)j * nkeep + idx[h]) * n31, p->raw + ((size_t)j * nkeep + first[np - 1]) * n31, n31)) same = 0;
               }
               if (same) w[np - 1] += cnt[keep[idx[h]]];
               else { first[np] = idx[h]; w[np] = cnt[keep[idx[h]]]; np++; }
               p->pose[idx[h]] = np - 1;
            }
         }
         else
            for (h = 0; h < nkeep; h++) { first[np] = h; w[np] = cnt[keep[h]]; p->pose[h] = np; np++; }
         p->n_pose = nkeep;
         /* com.posG / com.lgene: first pattern and number of sites of every gene (PatternWeight treesub.c:1428, 1466-1472) */
         p->posG[0] = 0; p->posG[1] = np; p->lgene[0] = nkeep;
         if (kg) {
            for (k = 0; k < p->ngene; k++) p->lgene[k] = 0;
            for (h = 0; h < nkeep; h++) p->lgene[kg[h]] += readpattern ? (int)(cnt[keep[h]] + 0.5) : 1;      /* sites of the gene */
            for (k = 0, h = 0; k < p->ngene; k++) {
               if (!p->lgene[k]) return pamlh_fail(p, "gene %d does not have any sites", k + 1);
               p->posG[k] = h;
               while (h < np && kg[first[h]] == k) h++;
            }
            p->posG[p->ngene] = np;
            free(kg);
         }
         raw2 = (char *)malloc((size_t)ns * np * n31);
         for (j = 0; j < ns; j++)
            for (h = 0; h < np; h++) memcpy(raw2 + ((size_t)j * np + h) * n31, p->raw + ((size_t)j * nkeep + first[h]) * n31, n31);
         free(p->raw);
         p->raw = raw2;
         p->npatt = np;
         p->w = (double *)malloc(np * sizeof(double));
         memcpy(p->w, w, np * sizeof(double));
         if (readpattern) {      /* com.ls = sum of counts when they exceed 1 (treesub.c:964-967) */
            double s = 0;
            for (h = 0; h < np; h++) s += w[h];
            if (s > 1.00001) p->ls = (int)(s + 0.5);
         }
         free(first); free(w);
      }
      free(idx); free(keep); free(cnt);
   }
   free(seq); free(line); free(site_gene); free(seqmem);

   /* encode (EncodeSeqs / SetMapAmbiguity) */
   {
      const int np = p->npatt, n = p->n;
      p->z = (unsigned char *)malloc((size_t)ns * np);
      if (p->seqtype != 1) {
         const int ncodes_all = (int)strlen(alpha);
         p->n_codes = p->cleandata ? n : ncodes_all;
         p->n_chara = (int *)calloc(p->n_codes, sizeof(int));
         p->chara_map = (unsigned char *)calloc((size_t)p->n_codes * n, 1);
         for (i = 0; i < p->n_codes; i++) {
            if (i < n) { p->n_chara[i] = 1; p->chara_map[(size_t)i * n] = (unsigned char)i; }
            else if (p->seqtype == 0) {
               int set[4], m = base_set(BASEs[i], set);
               p->n_chara[i] = m;
               for (k = 0; k < m; k++) p->chara_map[(size_t)i * n + k] = (unsigned char)set[k];
            }
            else { p->n_chara[i] = n; for (k = 0; k < n; k++) p->chara_map[(size_t)i * n + k] = (unsigned char)k; }
         }
         for (j = 0; j < ns; j++)
            for (h = 0; h < np; h++) p->z[(size_t)j * np + h] = (unsigned char)(strchr(alpha, p->raw[(size_t)j * np + h]) - alpha);
      }
      else {
         int from64[64], nsense = 0, namb = 0;
         char amb[256][4];
         for (i = 0; i < 64; i++) from64[i] = p->code[i] == '*' ? -1 : nsense++;
         for (j = 0; j < ns; j++)
            for (h = 0; h < np; h++) {
               const char *c = p->raw + ((size_t)j * np + h) * 3;
               int b[3], code;
               for (k = 0; k < 3; k++) b[k] = (int)(strchr(BASEs, c[k]) - BASEs);
               if (b[0] < 4 && b[1] < 4 && b[2] < 4) {
                  code = from64[b[0] * 16 + b[1] * 4 + b[2]];
                  if (code < 0) return pamlh_fail(p, "stop codon %.3s in sequence %d", c, j + 1);
               }
               else {
                  for (k = 0; k < namb; k++)
                     if (!memcmp(amb[k], c, 3)) break;
                  if (k == namb) {
                     if (namb >= 256 - n) return pamlh_fail(p, "too many distinct ambiguous codons");
                     memcpy(amb[namb], c, 3); amb[namb][3] = 0; namb++;
                  }
                  code = n + k;
               }
               p->z[(size_t)j * np + h] = (unsigned char)code;
            }
         p->n_codes = n + namb;
         p->n_chara = (int *)calloc(p->n_codes, sizeof(int));
         p->chara_map = (unsigned char *)calloc((size_t)p->n_codes * n, 1);
         for (i = 0; i < n; i++) { p->n_chara[i] = 1; p->chara_map[(size_t)i * n] = (unsigned char)i; }
         for (i = 0; i < namb; i++) {
            int s0[4], s1[4], s2[4], n0 = base_set(amb[i][0], s0), n1 = base_set(amb[i][1], s1), n2 = base_set(amb[i][2], s2);
            int i0, i1, i2, m = 0;
            for (i0 = 0; i0 < n0; i0++)
               for (i1 = 0; i1 < n1; i1++)
                  for (i2 = 0; i2 < n2; i2++) {
                     int ic = s0[i0] * 16 + s1[i1] * 4 + s2[i2];
                     if (from64[ic] >= 0) p->chara_map[(size_t)(n + i) * n + m++] = (unsigned char)from64[ic];
                  }
            if (!m) return pamlh_fail(p, "codon %s is a stop codon", amb[i]);
            p->n_chara[n + i] = m;
         }
         if (namb == 0) p->cleandata = 1;
      }
   }
   return 0;
}

/* ------------------------------------------------------------------ tree */
int pamlh_read_tree(pamlh *p)
{
   FILE *f = fopen(p->treefile, "r");
   char *buf, *s;
   long len;
   int ns = p->ns, nn = 0, i, depth = 0, cur = -1, nb = 0, maxn = 2 * p->ns;
   int *stack, *father, *nson, *sonbuf, last = -1, *label2, any_clade = 0;
   unsigned char *labelled;
   if (!f) return pamlh_fail(p, "cannot open tree file %s", p->treefile);
   fseek(f, 0, SEEK_END); len = ftell(f); fseek(f, 0, SEEK_SET);
   buf = (char *)malloc(len + 1);
   len = (long)fread(buf, 1, len, f); buf[len] = 0;
   fclose(f);
   s = strchr(buf, '(');
   if (!s) { free(buf); return pamlh_fail(p, "no tree in %s", p->treefile); }
   {      /* trees end with ';' (the last one may lack it) */
      char *q = s;
      p->ntrees = 0;
      while (q && *q) { char *e = strchr(q, ';'); p->ntrees++; q = e ? strchr(e + 1, '(') : NULL; }
      {      /* an "ns ntree" line in front of the trees limits the count (the reference reads that many: GetTreeFileType, codeml.c:617) */
         int hns = 0, hnt = 0;
         char c = *s;
         *s = 0;
         const int nh = sscanf(buf, "%d%d", &hns, &hnt);      /* "ns ntree" (paml), or "ntree" alone (phylip / molphy style) */
         if (nh == 1) hnt = hns;
         if (nh >= 1 && hnt >= 1 && hnt < p->ntrees) p->ntrees = hnt;
         *s = c;
      }
      if (p->itree < 0 || p->itree >= p->ntrees) { free(buf); return pamlh_fail(p, "tree %d asked for, %s holds %d", p->itree + 1, p->treefile, p->ntrees); }
      for (i = 0; i < p->itree; i++) s = strchr(strchr(s, ';') + 1, '(');
   }
   stack = (int *)malloc(maxn * sizeof(int));
   father = (int *)malloc(maxn * sizeof(int));
   p->label = (int *)calloc(maxn, sizeof(int));
   p->tree_branch = (double *)malloc(maxn * sizeof(double));
   p->branch_node = (int *)malloc(maxn * sizeof(int));
   label2 = (int *)malloc(maxn * sizeof(int)); labelled = (unsigned char *)calloc(maxn, 1);
   for (i = 0; i < maxn; i++) { father[i] = -1; p->tree_branch[i] = -1; label2[i] = -1; }
   nn = ns;
   for (; *s && *s != ';'; ) {
      if (isspace((unsigned char)*s)) { s++; continue; }
      if (*s == '(') {
         int node = nn++;
         if (nn > maxn) { free(buf); return pamlh_fail(p, "tree has too many nodes"); }
         if (depth > 0) { father[node] = stack[depth - 1]; p->branch_node[nb++] = node; }
         else p->root = node;
         stack[depth++] = node;
         s++;
         last = -1;
      }
      else if (*s == ',') { s++; last = -1; }
      else if (*s == ')') { last = stack[--depth]; s++; }
      else if (*s == ':' ) { char *e; double v = strtod(s + 1, &e); if (last >= 0) p->tree_branch[last] = v; s = e; }
      else if (*s == '#' || *s == '$') {      /* '#k': the label of this branch; '$k': of every branch of the clade without a label of its own */
         char *e; double v = strtod(s + 1, &e);
         if (last >= 0 && *s == '#') { p->label[last] = (int)v; labelled[last] = 1; }
         else if (last >= 0) { label2[last] = (int)v; any_clade = 1; }
         s = e;
      }
      else if (*s == '[') { while (*s && *s != ']') s++; if (*s) s++; }
      else if (last >= 0 && cur != last && (isalnum((unsigned char)*s) || *s == '_' || *s == '.') && s[-1] == ')') {
         while (*s && !strchr(",():;#$[ \t\r\n", *s)) s++;    /* internal node name / support value: ignored */
      }
      else {   /* a tip: name or 1-based number */
         char name[128];
         int k = 0, tip = -1;
         while (*s && !strchr(",():;#$[", *s) && k < 127) name[k++] = *s++;
         while (k > 0 && isspace((unsigned char)name[k - 1])) k--;
         name[k] = 0;
         for (i = 0; i < ns; i++)
            if (!strcmp(name, p->names[i])) { tip = i; break; }
         if (tip < 0) {
            char *e; long v = strtol(name, &e, 10);
            if (*e == 0 && v >= 1 && v <= ns) tip = (int)v - 1;
         }
         if (tip < 0) { free(buf); return pamlh_fail(p, "species %s in the tree is not in the sequence file", name); }
         if (depth < 1) { free(buf); return pamlh_fail(p, "bad tree"); }
         father[tip] = stack[depth - 1];
         p->branch_node[nb++] = tip;
         last = tip;
      }
   }
   free(buf);
   p->nnode = nn; p->nbranch = nb;
   nson = (int *)calloc(nn, sizeof(int));
   for (i = 0; i < nb; i++) nson[father[p->branch_node[i]]]++;
   p->sons_ptr = (int *)calloc(nn + 1, sizeof(int));
   for (i = 0; i < nn; i++) p->sons_ptr[i + 1] = p->sons_ptr[i] + nson[i];
   sonbuf = (int *)malloc((nb + 1) * sizeof(int));
   memset(nson, 0, nn * sizeof(int));
   for (i = 0; i < nb; i++) { int c = p->branch_node[i], fa = father[c]; sonbuf[p->sons_ptr[fa] + nson[fa]++] = c; }   /* appearance order = sons[] order */
   p->sons = sonbuf;
   p->father = father;
   if (any_clade) {      /* clade labels (DownTreeCladeLabel treesub.c:2960-2976): walking down from the root with label 0, a '$k' node switches the
                          * current label to k for its whole clade; every branch without a '#' of its own takes the current label */
      int sp = 0, *cur = (int *)malloc(nn * sizeof(int));
      stack[sp] = p->root; cur[sp++] = 0;
      while (sp) {
         const int x = stack[--sp], lab = label2[x] != -1 ? label2[x] : cur[sp];
         int j;
         if (x != p->root && !labelled[x]) p->label[x] = lab;
         for (j = p->sons_ptr[x]; j < p->sons_ptr[x + 1]; j++) { stack[sp] = p->sons[j]; cur[sp++] = lab; }
      }
      free(cur);
   }
   free(label2); free(labelled);
   /* SetNodeScale (treesub.c:7177-7197) */
   {
      const int every = p->is_codeml ? (p->seqtype == 1 ? 15 : 50) : 100;
      int *cnt = (int *)calloc(nn, sizeof(int)), *order = (int *)malloc(nn * sizeof(int)), no = 0, sp = 0;
      p->scale = (unsigned char *)calloc(nn, 1);
      stack[sp++] = p->root;     /* post-order via reversed pre-order */
      while (sp) { int x = stack[--sp], j; order[no++] = x; for (j = p->sons_ptr[x]; j < p->sons_ptr[x + 1]; j++) stack[sp++] = p->sons[j]; }
      for (i = no - 1; i >= 0; i--) {
         int x = order[i], j, d = 0;
         if (p->sons_ptr[x + 1] == p->sons_ptr[x]) continue;
         for (j = p->sons_ptr[x]; j < p->sons_ptr[x + 1]; j++) { int c = p->sons[j]; d += (p->sons_ptr[c + 1] > p->sons_ptr[c]) ? cnt[c] : 1; }
         if (x != p->root && d > every) { p->scale[x] = 1; d = 1; }
         cnt[x] = d;
      }
      free(cnt); free(order);
   }
   free(stack); free(nson);
   return 0;
}
