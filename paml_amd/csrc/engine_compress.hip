// engine_compress.hip — site-pattern compression (PatternWeight treesub.c:1386) on the device; kernels in compress.h.
// Stand-alone: no engine.  Built for gfx950 only.
#include <hip/hip_runtime.h>

#include "../../include/paml_amd.h"
#include "compress.h"

using namespace paml_amd;

extern "C" int paml_amd_compress_patterns(int n_seq, int n_sites, int width, const unsigned char *chars, const int *gene, int *n_patt,
                                          int *first_site, double *weights, int *pose)
{
   if (n_seq < 1 || n_sites < 1 || width < 1 || !chars || !n_patt || !first_site || !weights || !pose) return PAML_AMD_EINVAL;
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return PAML_AMD_EHIP;
   const int n = n_sites, n_tiles = (n + CMP_TILE - 1) / CMP_TILE, nb = (n + CMP_THREADS - 1) / CMP_THREADS;
   const size_t row = (size_t)n * width;
   unsigned char *d_chars = nullptr, *d_dig = nullptr;
   int *d_gene = nullptr, *d_idx[2] = {nullptr, nullptr}, *d_hist = nullptr, *d_head = nullptr, *d_sums = nullptr, *d_pose = nullptr, *d_first = nullptr,
       *d_start = nullptr, *d_total = nullptr, *d_rowsum = nullptr;
   double *d_w = nullptr;
   int rc = 0, total = 0, cur = 0;
#define CMPCHK(call) do { if ((call) != hipSuccess) { rc = PAML_AMD_EHIP; goto done; } } while (0)
   CMPCHK(hipMalloc(&d_chars, row * n_seq)); CMPCHK(hipMalloc(&d_dig, (size_t)n));
   CMPCHK(hipMalloc(&d_idx[0], (size_t)n * 4)); CMPCHK(hipMalloc(&d_idx[1], (size_t)n * 4));
   CMPCHK(hipMalloc(&d_hist, (size_t)256 * n_tiles * 4));
   CMPCHK(hipMalloc(&d_head, (size_t)n * 4)); CMPCHK(hipMalloc(&d_sums, (size_t)n_tiles * 4));
   CMPCHK(hipMalloc(&d_pose, (size_t)n * 4)); CMPCHK(hipMalloc(&d_first, (size_t)n * 4)); CMPCHK(hipMalloc(&d_start, (size_t)n * 4));
   CMPCHK(hipMalloc(&d_total, 4)); CMPCHK(hipMalloc(&d_w, (size_t)n * 8)); CMPCHK(hipMalloc(&d_rowsum, 256 * 4));
   CMPCHK(hipMemcpy(d_chars, chars, row * n_seq, hipMemcpyHostToDevice));
   if (gene) { CMPCHK(hipMalloc(&d_gene, (size_t)n * 4)); CMPCHK(hipMemcpy(d_gene, gene, (size_t)n * 4, hipMemcpyHostToDevice)); }
   {
      CompressArgs a{};
      a.n_sites = n; a.n_seq = n_seq; a.width = width; a.n_tiles = n_tiles; a.row_stride = (long)row; a.chars = d_chars; a.gene = d_gene; a.hist = d_hist; a.dig = d_dig;
      hipLaunchKernelGGL(cmp_iota, dim3(nb), dim3(CMP_THREADS), 0, 0, d_idx[0], n);
      // least significant key byte first: the last character of the last sequence ... the first of the first, then the gene
      for (int kb = n_seq * width - 1; kb >= (gene ? -1 : 0); kb--) {
         a.seq = kb < 0 ? -1 : kb / width; a.pos = kb < 0 ? 0 : kb % width;
         a.idx_in = d_idx[cur]; a.idx_out = d_idx[cur ^ 1];
         hipLaunchKernelGGL(cmp_hist, dim3(n_tiles), dim3(CMP_THREADS), 0, 0, a);
         hipLaunchKernelGGL(cmp_row_sums, dim3(256), dim3(CMP_THREADS), 0, 0, d_hist, n_tiles, d_rowsum);
         hipLaunchKernelGGL(cmp_row_scan, dim3(256), dim3(CMP_THREADS), 0, 0, d_hist, n_tiles, d_rowsum);
         hipLaunchKernelGGL(cmp_scatter, dim3(n_tiles), dim3(CMP_THREADS), 0, 0, a);
         cur ^= 1;
      }
      a.idx_in = d_idx[cur];
      hipLaunchKernelGGL(cmp_heads, dim3(nb), dim3(CMP_THREADS), 0, 0, a, d_head);
      hipLaunchKernelGGL(cmp_tile_sums, dim3(n_tiles), dim3(CMP_THREADS), 0, 0, d_head, n, d_sums);
      hipLaunchKernelGGL(cmp_scan1, dim3(1), dim3(1024), 0, 0, d_sums, (long)n_tiles, d_total);
      hipLaunchKernelGGL(cmp_number, dim3(n_tiles), dim3(CMP_THREADS), 0, 0, d_head, d_sums, d_idx[cur], n, d_pose, d_first, d_start);
      CMPCHK(hipGetLastError());
      CMPCHK(hipMemcpy(&total, d_total, 4, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL(cmp_weights, dim3((total + CMP_THREADS - 1) / CMP_THREADS), dim3(CMP_THREADS), 0, 0, d_start, total, n, d_w);
      CMPCHK(hipGetLastError());
   }
   CMPCHK(hipMemcpy(pose, d_pose, (size_t)n * 4, hipMemcpyDeviceToHost));
   CMPCHK(hipMemcpy(first_site, d_first, (size_t)total * 4, hipMemcpyDeviceToHost));
   CMPCHK(hipMemcpy(weights, d_w, (size_t)total * 8, hipMemcpyDeviceToHost));
   *n_patt = total;
done:
#undef CMPCHK
   {
      void *bufs[] = {d_chars, d_gene, d_idx[0], d_idx[1], d_hist, d_head, d_sums, d_pose, d_first, d_start, d_total, d_w, d_rowsum, d_dig};
      for (void *b : bufs) (void)hipFree(b);
   }
   return rc;
}
