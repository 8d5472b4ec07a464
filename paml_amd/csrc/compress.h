// Site-pattern compression on the device (PatternWeight, treesub.c:1386-1516).
//
// The reference collapses the ls alignment columns into sorted distinct patterns by binary-search insertion into a sorted index
// (O(ls log npatt) string compares + up to O(npatt^2) index moves).  Here the columns are SORTED: a least-significant-digit radix
// sort of the site indices over the key bytes of a column (sequence-major: all characters of a site in sequence 0, then
// sequence 1, ...; optionally the gene id as the most significant digit), one stable counting-sort pass per byte position.
// Order = memcmp order of the raw characters = the reference's strcmp order of its (char + 1) strings.  Because every pass is
// stable and the start is the identity, equal columns stay in site order, so the first site of a run is the one the reference
// keeps (p2s[]).  Then one pass marks run heads, a scan numbers the runs, and pose / first_site / weights fall out.
//
// This is byte work bound by memory latency / bandwidth (random 1-byte gathers through the permutation); nothing here wants
// the matrix cores.  Per pass: histogram per 2048-element tile (LDS atomics), one scan of the [digit][tile] table, a stable
// scatter (wave ballots rank equal digits inside a wave, LDS counters across the 4 waves and across the tile's rounds).
#pragma once
#include <hip/hip_runtime.h>

namespace paml_amd {

#define CMP_TILE 2048
#define CMP_THREADS 256

struct CompressArgs {
   int n_sites, n_seq, width, n_tiles;
   long row_stride;               // bytes between sequences = n_sites * width
   const unsigned char *chars;    // [n_seq][n_sites * width]
   const int *gene;               // [n_sites] or null
   const int *idx_in;
   int *idx_out;
   int *hist;                     // [256][n_tiles]
   unsigned char *dig;            // [n_sites] digit of every sorted position, left by the histogram pass for the scatter
   int seq, pos;                  // key byte of this pass: chars[seq][site * width + pos]; seq < 0: the gene id
};

__device__ __forceinline__ int cmp_digit(const CompressArgs &a, int site)
{
   if (a.seq < 0) return a.gene[site] & 255;
   return a.chars[(long)a.seq * a.row_stride + (long)site * a.width + a.pos];
}

__global__ __launch_bounds__(CMP_THREADS) void cmp_iota(int *idx, int n)
{
   const int i = blockIdx.x * CMP_THREADS + threadIdx.x;
   if (i < n) idx[i] = i;
}

__global__ __launch_bounds__(CMP_THREADS) void cmp_hist(CompressArgs a)
{
   __shared__ int h[256];
   h[threadIdx.x] = 0;
   __syncthreads();
   const int base = blockIdx.x * CMP_TILE;
   for (int r = 0; r < CMP_TILE / CMP_THREADS; r++) {
      const int e = base + r * CMP_THREADS + threadIdx.x;
      if (e < a.n_sites) {
         const int d = cmp_digit(a, a.idx_in[e]);      // the random gather of the pass: done once, kept for the scatter
         a.dig[e] = (unsigned char)d;
         atomicAdd(&h[d], 1);
      }
   }
   __syncthreads();
   a.hist[(long)threadIdx.x * a.n_tiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of m ints in place by ONE block of 1024 threads (m = 256 * n_tiles, or the flags' tile sums)
__global__ __launch_bounds__(1024) void cmp_scan1(int *v, long m, int *total)
{
   __shared__ long s[1024];
   const long per = (m + 1023) / 1024, lo = (long)threadIdx.x * per, hi = lo + per < m ? lo + per : m;
   long sum = 0;
   for (long i = lo; i < hi; i++) sum += v[i];
   s[threadIdx.x] = sum;
   __syncthreads();
   for (int off = 1; off < 1024; off <<= 1) {
      const long t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
   }
   long run = s[threadIdx.x] - sum;
   for (long i = lo; i < hi; i++) { const int x = v[i]; v[i] = (int)run; run += x; }
   if (total && threadIdx.x == 1023) *total = (int)s[1023];
}

// The [digit][tile] table of one pass has 256 rows of n_tiles counts.  Its exclusive scan in row-major order, coalesced:
// row sums (one block per digit), a scan of the 256 sums (one wave-sized job), then every row scanned with its base.
__global__ __launch_bounds__(CMP_THREADS) void cmp_row_sums(const int *hist, int n_tiles, int *rowsum)
{
   __shared__ int s[CMP_THREADS];
   const int *row = hist + (long)blockIdx.x * n_tiles;
   int t = 0;
   for (int i = threadIdx.x; i < n_tiles; i += CMP_THREADS) t += row[i];
   s[threadIdx.x] = t;
   __syncthreads();
   for (int st = CMP_THREADS / 2; st >= 1; st >>= 1) {
      if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
      __syncthreads();
   }
   if (threadIdx.x == 0) rowsum[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(CMP_THREADS) void cmp_row_scan(int *hist, int n_tiles, const int *rowsum)
{
   __shared__ int s[CMP_THREADS], carry;
   // base of this digit = sum of the rows before it (256 values: every block redoes this small sum)
   int b = 0;
   for (int d = threadIdx.x; d < (int)blockIdx.x; d += CMP_THREADS) b += rowsum[d];
   s[threadIdx.x] = b;
   __syncthreads();
   for (int st = CMP_THREADS / 2; st >= 1; st >>= 1) {
      if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
      __syncthreads();
   }
   if (threadIdx.x == 0) carry = s[0];
   __syncthreads();
   int *row = hist + (long)blockIdx.x * n_tiles;
   for (int i0 = 0; i0 < n_tiles; i0 += CMP_THREADS) {
      const int i = i0 + threadIdx.x, v = i < n_tiles ? row[i] : 0;
      s[threadIdx.x] = v;
      __syncthreads();
      for (int off = 1; off < CMP_THREADS; off <<= 1) {
         const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
         __syncthreads();
         s[threadIdx.x] += t;
         __syncthreads();
      }
      if (i < n_tiles) row[i] = carry + s[threadIdx.x] - v;
      __syncthreads();
      if (threadIdx.x == 0) carry += s[CMP_THREADS - 1];
      __syncthreads();
   }
}

__global__ __launch_bounds__(CMP_THREADS) void cmp_scatter(CompressArgs a)
{
   __shared__ int base[256], wcnt[4][256];
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   base[threadIdx.x] = a.hist[(long)threadIdx.x * a.n_tiles + blockIdx.x];
   const int t0 = blockIdx.x * CMP_TILE;
   for (int r = 0; r < CMP_TILE / CMP_THREADS; r++) {
      for (int w = 0; w < 4; w++) wcnt[w][threadIdx.x] = 0;
      __syncthreads();
      const int e = t0 + r * CMP_THREADS + threadIdx.x;
      const bool valid = e < a.n_sites;
      const int site = valid ? a.idx_in[e] : 0;
      const int d = valid ? a.dig[e] : 0;
      // lanes of this wave holding the same digit
      unsigned long long m = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; bit++) {
         const unsigned long long b = __ballot((d >> bit) & 1);
         m &= ((d >> bit) & 1) ? b : ~b;
      }
      const int rank = __popcll(m & ((1ull << lane) - 1ull));
      if (valid && rank == 0) wcnt[wave][d] = __popcll(m);
      __syncthreads();
      if (valid) {
         int pre = 0;
         for (int w = 0; w < wave; w++) pre += wcnt[w][d];
         a.idx_out[base[d] + pre + rank] = site;
      }
      __syncthreads();
      base[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
      __syncthreads();
   }
}

// head[i] = 1 when sorted position i starts a new pattern (different gene or different column than position i - 1)
__global__ __launch_bounds__(CMP_THREADS) void cmp_heads(CompressArgs a, int *head)
{
   const int i = blockIdx.x * CMP_THREADS + threadIdx.x;
   if (i >= a.n_sites) return;
   int differ = i == 0;
   if (!differ) {
      const int s1 = a.idx_in[i], s0 = a.idx_in[i - 1];
      if (a.gene && a.gene[s1] != a.gene[s0]) differ = 1;
      for (int j = 0; j < a.n_seq && !differ; j++) {
         const unsigned char *row = a.chars + (long)j * a.row_stride;
         for (int c = 0; c < a.width; c++)
            if (row[(long)s1 * a.width + c] != row[(long)s0 * a.width + c]) { differ = 1; break; }
      }
   }
   head[i] = differ;
}

// per-tile sums of head[] (for the two-level scan)
__global__ __launch_bounds__(CMP_THREADS) void cmp_tile_sums(const int *head, int n, int *sums)
{
   __shared__ int s[CMP_THREADS];
   int t = 0;
   const int base = blockIdx.x * CMP_TILE;
   for (int r = 0; r < CMP_TILE / CMP_THREADS; r++) {
      const int e = base + r * CMP_THREADS + threadIdx.x;
      if (e < n) t += head[e];
   }
   s[threadIdx.x] = t;
   __syncthreads();
   for (int st = CMP_THREADS / 2; st >= 1; st >>= 1) {
      if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
      __syncthreads();
   }
   if (threadIdx.x == 0) sums[blockIdx.x] = s[0];
}

// pattern number of every sorted position (inclusive scan of head - 1) -> pose[site], first_site[pattern], start[pattern]
__global__ __launch_bounds__(CMP_THREADS) void cmp_number(const int *head, const int *tile_off, const int *idx, int n, int *pose, int *first_site,
                                                          int *start)
{
   __shared__ int s[CMP_THREADS];
   const int base = blockIdx.x * CMP_TILE + threadIdx.x * (CMP_TILE / CMP_THREADS);     // 8 consecutive positions per thread
   int loc[CMP_TILE / CMP_THREADS], t = 0;
   for (int k = 0; k < CMP_TILE / CMP_THREADS; k++) { loc[k] = base + k < n ? head[base + k] : 0; t += loc[k]; }
   s[threadIdx.x] = t;
   __syncthreads();
   for (int off = 1; off < CMP_THREADS; off <<= 1) {
      const int v = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += v;
      __syncthreads();
   }
   int run = tile_off[blockIdx.x] + s[threadIdx.x] - t;       // heads before this thread's first position
   for (int k = 0; k < CMP_TILE / CMP_THREADS; k++) {
      const int i = base + k;
      if (i >= n) break;
      run += loc[k];
      const int pid = run - 1, site = idx[i];
      pose[site] = pid;
      if (loc[k]) { first_site[pid] = site; start[pid] = i; }
   }
}

__global__ __launch_bounds__(CMP_THREADS) void cmp_weights(const int *start, int n_patt, int n, double *w)
{
   const int p = blockIdx.x * CMP_THREADS + threadIdx.x;
   if (p < n_patt) w[p] = (double)((p + 1 < n_patt ? start[p + 1] : n) - start[p]);
}

}  // namespace paml_amd
