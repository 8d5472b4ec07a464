// engine_comm.hip — site patterns sharded over several GPUs: shard bounds, the RCCL communicator of the ranks' engines.
// Built for gfx950 only (one of the translation units of libpaml_amd.so, see engine_state.h).
#include "engine_state.h"

extern "C" {

// ---- pattern shards over several GPUs ---------------------------------------------------------------------------------
int paml_amd_device_count(void)
{
   int n = 0;
   return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int paml_amd_set_device(int device) { return hipSetDevice(device) == hipSuccess ? 0 : PAML_AMD_EHIP; }

int paml_amd_shard_bounds(long n_patt_global, int world, int rank, long *first, long *count)
{
   if (n_patt_global < 1 || world < 1 || rank < 0 || rank >= world || !first || !count) return PAML_AMD_EINVAL;
   const long chunk = red_chunk(n_patt_global), nb = (n_patt_global + chunk - 1) / chunk;
   // Fewer reduction chunks than ranks would leave ranks without patterns: refused — for EVERY rank alike (the answer depends on
   // n_patt_global and world only), so that all ranks of a job fail together before any of them enters a collective call.
   if (world > nb) return PAML_AMD_EUNSUPPORTED;
   const long c0 = nb * rank / world, c1 = nb * (rank + 1) / world;      // chunks [c0, c1): as even as whole chunks allow
   *first = std::min(n_patt_global, c0 * chunk);
   *count = std::min(n_patt_global, c1 * chunk) - *first;
   return 0;
}

int paml_amd_max_ranks(long n_patt_global)
{
   if (n_patt_global < 1) return 0;
   const long chunk = red_chunk(n_patt_global);
   return (int)std::min<long>((n_patt_global + chunk - 1) / chunk, 1 << 20);
}

int paml_amd_comm_unique_id(void *id128)
{
   if (!id128) return PAML_AMD_EINVAL;
   static_assert(sizeof(ncclUniqueId) == PAML_AMD_COMM_ID_BYTES, "paml_amd.h states the size of ncclUniqueId");
   if (!rccl().load()) return PAML_AMD_EUNSUPPORTED;
   ncclUniqueId id;
   if (rccl().GetUniqueId(&id) != ncclSuccess) return PAML_AMD_EHIP;
   memcpy(id128, &id, sizeof(id));
   return 0;
}

int paml_amd_comm_init(paml_amd_engine *e, int rank, int world, const void *id128, long n_patt_global, long first_pattern)
{
   enter(e);
   if (!e || world < 1 || rank < 0 || rank >= world || (world > 1 && !id128)) return fail(e, PAML_AMD_EINVAL, "comm_init: bad arguments");
   if (e->comm) return fail(e, PAML_AMD_EINVAL, "comm_init: the engine already has a communicator");
   if (e->small20 && n_patt_global != e->n_patt)
      return fail(e, PAML_AMD_EINVAL, "comm_init: this 20-state engine chose its kernels by its own (small) size; create the engines of shards with PAML_AMD_SHARD");
   const int chunk = red_chunk(n_patt_global);
   if (first_pattern < 0 || first_pattern + e->n_patt > n_patt_global || first_pattern % chunk != 0 ||
       (first_pattern + e->n_patt != n_patt_global && e->n_patt % chunk != 0))
      return fail(e, PAML_AMD_EINVAL, "comm_init: the shard must start and (unless it is the last) end at multiples of " +
                                         std::to_string(chunk) + " patterns (paml_amd_shard_bounds)");
   HIPCHK(hipStreamSynchronize(e->stream));
   if (id128) {      // world == 1 with an id: a one-rank communicator (exercises the collective path on a single GPU)
      if (!rccl().load()) return fail(e, PAML_AMD_EUNSUPPORTED, "comm_init: " + rccl().err);
      ncclUniqueId id;
      memcpy(&id, id128, sizeof(id));
      const ncclResult_t nr = rccl().CommInitRank(&e->comm, world, id, rank);
      if (nr != ncclSuccess) {
         e->comm = nullptr;
         return fail(e, PAML_AMD_EHIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(nr));
      }
   }
   if (e->comm)
      if (int rc = ensure_side_stream(e)) return rc;
   if (e->sc) HIPCHK(hipStreamSynchronize(e->sc));
   for (bool &b : e->done_pending) b = false;
   for (bool &b : e->join_pending) b = false;
   e->red_slot = e->last_slot = 0;
   e->rank = rank; e->world = world;
   e->n_patt_global = n_patt_global; e->first_patt = first_pattern;
   e->chunk = chunk;
   e->nb_global = (int)((n_patt_global + chunk - 1) / chunk);
   e->first_chunk = (int)(first_pattern / chunk);
   e->d_partial.release();      // re-zeroed at their new size by the next evaluations
   for (auto &b : e->d_partial_s) b.release();
   return 0;
}

int paml_amd_comm_destroy(paml_amd_engine *e)
{
   enter(e);
   if (!e) return PAML_AMD_EINVAL;
   HIPCHK(hipStreamSynchronize(e->stream));
   if (e->sc) HIPCHK(hipStreamSynchronize(e->sc));
   if (e->comm) (void)rccl().CommDestroy(e->comm);
   e->comm = nullptr;
   e->red_slot = e->last_slot = 0;
   e->rank = 0; e->world = 1; e->n_patt_global = e->n_patt; e->first_patt = 0;
   e->chunk = red_chunk(e->n_patt); e->nb_global = (e->n_patt + e->chunk - 1) / e->chunk; e->first_chunk = 0;
   e->d_partial.release();
   for (auto &b : e->d_partial_s) b.release();
   return 0;
}

int paml_amd_get_partial_sums(paml_amd_engine *e, double *out, int cap)
{
   enter(e);
   if (!e || !out) return fail(e, PAML_AMD_EINVAL, "get_partial_sums: null argument");
   if (e->n_eval == 0 || !e->part_slot(e->last_slot).p || cap < e->nb_global) return fail(e, PAML_AMD_EINVAL, "get_partial_sums: nothing evaluated yet, or cap < number of chunks");
   const double *src = e->comm ? e->tot_slot(e->last_slot).p : e->part_slot(e->last_slot).p;
   HIPCHK(hipMemcpyAsync(out, src, (size_t)e->nb_global * sizeof(double), hipMemcpyDeviceToHost, e->stream));
   HIPCHK(hipStreamSynchronize(e->stream));
   return e->nb_global;
}

int paml_amd_comm_library(char *path, int cap)
{
   if (!path || cap < 1) return PAML_AMD_EINVAL;
   if (!rccl().load()) return PAML_AMD_EUNSUPPORTED;
   Dl_info info;
   const char *nm = (dladdr((const void *)rccl().AllReduce, &info) && info.dli_fname) ? info.dli_fname : "?";
   const int len = (int)std::min<size_t>(strlen(nm), (size_t)cap - 1);
   memcpy(path, nm, len);
   path[len] = 0;
   return len;
}

int paml_amd_comm_stats(paml_amd_engine *e, int enable, int *n, double *ex_mean, double *ex_max, double *lw_mean, double *lw_max)
{
   if (!e) return PAML_AMD_EINVAL;
   if (n && ex_mean && ex_max && lw_mean && lw_max) {
      const int cnt = (int)std::min<long>(e->st_count, paml_amd_engine::NSTAT);
      double se = 0, me = 0, sw = 0, mw = 0;
      int got = 0;
      for (int k = 0; k < cnt; k++) {
         const int i = (int)((e->st_count - 1 - k) % paml_amd_engine::NSTAT);
         float ms = 0;
         if (hipEventElapsedTime(&ms, e->st_part[i], e->st_done[i]) != hipSuccess) { (void)hipGetLastError(); continue; }
         double w = 0;
         if (e->st_waited[i]) {
            float wm = 0;
            if (hipEventElapsedTime(&wm, e->st_w0[i], e->st_w1[i]) == hipSuccess) w = wm * 1e3;
            else (void)hipGetLastError();
         }
         se += ms * 1e3; me = std::max(me, (double)ms * 1e3); sw += w; mw = std::max(mw, w);
         got++;
      }
      *n = got;
      *ex_mean = got ? se / got : 0; *ex_max = me; *lw_mean = got ? sw / got : 0; *lw_max = mw;
   }
   e->comm_stats = enable != 0;
   if (!enable) e->st_count = 0;
   return 0;
}

int paml_amd_comm_info(const paml_amd_engine *e, int *rank, int *world, long *n_patt_global, long *first_pattern, int *chunk)
{
   if (!e) return PAML_AMD_EINVAL;
   if (rank) *rank = e->rank;
   if (world) *world = e->world;
   if (n_patt_global) *n_patt_global = e->n_patt_global;
   if (first_pattern) *first_pattern = e->first_patt;
   if (chunk) *chunk = e->chunk;
   return 0;
}

}  // extern "C"
