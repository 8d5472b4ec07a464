// engine_beb.hip — the Bayes-empirical-Bayes grid integral (lfunNSsites_M2M8 codeml.c:6482-6580) over the class likelihoods the last
// evaluation left on the device.
// Built for gfx950 only (one of the translation units of libpaml_amd.so, see engine_state.h).
#include "engine_state.h"
#include "kernels_beb.h"

extern "C" {

// shared front half of the two BEB entry points: checks, uploads, and the scale / lnfx / finish kernels
static int beb_front(paml_amd_engine *e, const char *who, int n_grid, int n_cls, const double *pcl, const int *iw, size_t out_per_patt, BebArgs &a)
{
   if (e->mode != PAML_AMD_MODE_LFUNDG || e->n_eval == 0 || !e->d_fhK.p)
      return fail(e, PAML_AMD_EINVAL, std::string(who) + ": needs a previous evaluation in the lfundG class mode");
   const int K = e->K, np = e->n_patt;
   for (long i = 0; i < (long)n_grid * n_cls; i++)
      if (iw[i] < 0 || iw[i] >= K) return fail(e, PAML_AMD_EINVAL, std::string(who) + ": class index out of range");
   a = BebArgs{};
   a.n_patt = np; a.K = K; a.n_grid = n_grid; a.n_cls = n_cls;
   a.log_form = e->tree.n_scale > 0;      // root_value: with scaling nodes fhK = log f + scale factors
   a.patt_per_blk = 4096;
   a.n_pblk = (np + a.patt_per_blk - 1) / a.patt_per_blk;
   HIPCHK(e->d_beb_f.ensure((size_t)K * np));
   HIPCHK(e->d_beb_part.ensure((size_t)n_grid * a.n_pblk));
   HIPCHK(e->d_beb_g.ensure((size_t)2 * n_grid + K + 1));
   HIPCHK(e->d_beb_out.ensure(out_per_patt * np));
   HIPCHK(upload(e->d_beb_pcl, pcl, (size_t)n_grid * n_cls, e->stream));
   HIPCHK(upload(e->d_beb_iw, iw, (size_t)n_grid * n_cls, e->stream));
   // (the class likelihoods of the LAST evaluation: a run of eval_device calls may have left them in another slot than d_fhK)
   a.fhK = e->fhk_slot(e->last_fhk).p; a.weights = e->d_weights.p; a.f = e->d_beb_f.p; a.pcl = e->d_beb_pcl.p; a.iw = e->d_beb_iw.p;
   a.part = e->d_beb_part.p; a.lnfxs = e->d_beb_g.p; a.wg = e->d_beb_g.p + n_grid; a.fx = e->d_beb_g.p + 2 * n_grid;
   a.w_class = e->d_beb_g.p + 2 * n_grid + 1;
   a.pr_last = e->d_beb_out.p; a.mean_w = a.pr_last + np; a.sd_w = a.mean_w + np;
   hipLaunchKernelGGL(beb_scale, dim3((np + 255) / 256), dim3(256), 0, e->stream, a);
   hipLaunchKernelGGL(beb_lnfx, dim3(a.n_pblk, (n_grid + 63) / 64), dim3(256), 0, e->stream, a);
   if (e->comm && e->world > 1) {
      // Pattern shards: a grid point's log-likelihood is a sum over ALL patterns, so the shards' sums are added over the
      // ranks before the grid weights are formed; everything per pattern (beb_scale, beb_post) stays on the shard.  Every rank
      // calls with the same grid; each gets the posteriors of its own patterns and the same ln_fx.
      a.phase = 1;
      hipLaunchKernelGGL(beb_finish, dim3(1), dim3(256), 0, e->stream, a);
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventRecord(e->ev_part[0], e->stream));
      HIPCHK(hipStreamWaitEvent(e->sc, e->ev_part[0], 0));
      const ncclResult_t nr = rccl().AllReduce(a.lnfxs, a.lnfxs, (size_t)n_grid, ncclDouble, ncclSum, e->comm, e->sc);
      if (nr != ncclSuccess) return fail(e, PAML_AMD_EHIP, std::string(who) + ": ncclAllReduce: " + rccl().GetErrorString(nr));
      HIPCHK(hipEventRecord(e->ev_done[0], e->sc));
      HIPCHK(hipStreamWaitEvent(e->stream, e->ev_done[0], 0));
      a.phase = 2;
   }
   hipLaunchKernelGGL(beb_finish, dim3(1), dim3(256), 0, e->stream, a);
   return 0;
}

int paml_amd_beb_grid(paml_amd_engine *e, int n_grid, int n_cls, const double *pcl, const int *iw, const double *w_class,
                      double *ln_fx, double *pr_last, double *mean_w, double *sd_w)
{
   enter(e);
   if (!e || n_grid < 1 || n_cls < 1 || !pcl || !iw || !w_class || !pr_last || !mean_w || !sd_w)
      return fail(e, PAML_AMD_EINVAL, "beb_grid: bad arguments");
   if (e->K > BEB_MAXK) return fail(e, PAML_AMD_EUNSUPPORTED, "beb_grid: more than 32 classes");
   BebArgs a;
   const int np = e->n_patt;
   if (int rc = beb_front(e, "beb_grid", n_grid, n_cls, pcl, iw, 3, a)) return rc;
   HIPCHK(hipMemcpyAsync((double *)a.w_class, w_class, (size_t)e->K * sizeof(double), hipMemcpyHostToDevice, e->stream));
   hipLaunchKernelGGL(beb_post, dim3((np + 255) / 256), dim3(256), 0, e->stream, a);
   HIPCHK(hipGetLastError());
   HIPCHK(hipMemcpyAsync(pr_last, a.pr_last, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipMemcpyAsync(mean_w, a.mean_w, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipMemcpyAsync(sd_w, a.sd_w, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (ln_fx) HIPCHK(hipMemcpyAsync(ln_fx, a.fx, sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return eigen_fail_check(e);      // (the grid's class likelihoods come from eigen systems a set_eigen_qrev_batch may have left unconverged)
}

int paml_amd_beb_grid_classes(paml_amd_engine *e, int n_grid, int n_cls, const double *pcl, const int *iw, double *ln_fx, double *post)
{
   enter(e);
   if (!e || n_grid < 1 || n_cls < 1 || n_cls > BEB_MAXCLS || !pcl || !iw || !post)
      return fail(e, PAML_AMD_EINVAL, "beb_grid_classes: bad arguments (at most 8 mixture classes per grid point)");
   BebArgs a;
   const int np = e->n_patt;
   if (int rc = beb_front(e, "beb_grid_classes", n_grid, n_cls, pcl, iw, (size_t)n_cls, a)) return rc;
   hipLaunchKernelGGL(beb_post_classes, dim3((np + 255) / 256), dim3(256), 0, e->stream, a);
   HIPCHK(hipGetLastError());
   HIPCHK(hipMemcpyAsync(post, a.pr_last, (size_t)n_cls * np * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   if (ln_fx) HIPCHK(hipMemcpyAsync(ln_fx, a.fx, sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return eigen_fail_check(e);      // (the grid's class likelihoods come from eigen systems a set_eigen_qrev_batch may have left unconverged)
}

}  // extern "C"
