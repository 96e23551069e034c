// Built-in RCCL implementation of nkp_comm_ops (include/nkp.h): one communicator per process,
// one GPU per process, collectives enqueued on the solver's stream.  xGMI traffic classes:
//   allreduce  -- 1..m+2 doubles per Gram-Schmidt pass: pure latency, one call per pass
//   alltoallv  -- the halo rows of the latitude-band neighbours (grouped ncclSend/ncclRecv)
#include "../../include/nkp.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

namespace {
struct RcclCtx {
   ncclComm_t comm = nullptr;
   int rank = 0, nranks = 1;
   void *stage = nullptr;
   size_t stage_bytes = 0;
   bool failed = false;          // a call on this communicator failed: it is aborted at teardown, not destroyed
};

int ensure_stage (RcclCtx *c, size_t bytes)
{
   if (bytes <= c->stage_bytes) return 0;
   if (c->stage) (void) hipFree (c->stage);
   c->stage = nullptr;
   c->stage_bytes = 0;
   if (hipMalloc (&c->stage, bytes) != hipSuccess) return 1;
   c->stage_bytes = bytes;
   return 0;
}

int rccl_allreduce (void *ctx, void *dev_buf, int count, int op, void *stream)
{
   RcclCtx *c = (RcclCtx *) ctx;
   const bool bad = ncclAllReduce (dev_buf, dev_buf, (size_t) count, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, (hipStream_t) stream) != ncclSuccess;
   if (bad) c->failed = true;
   return bad;
}

template <class T>
int exchange (RcclCtx *c, const T *send, const int *scnt, T *recv, const int *rcnt, ncclDataType_t dt, hipStream_t st)
{
   ncclResult_t r = ncclGroupStart ();
   size_t so = 0, ro = 0;
   for (int p = 0; p < c->nranks && r == ncclSuccess; p++) {
      if (p != c->rank) {
         if (scnt[p]) r = ncclSend (send + so, (size_t) scnt[p], dt, p, c->comm, st);
         if (r == ncclSuccess && rcnt[p]) r = ncclRecv (recv + ro, (size_t) rcnt[p], dt, p, c->comm, st);
      } else if (scnt[p]) {
         if (hipMemcpyAsync (recv + ro, send + so, (size_t) scnt[p] * sizeof (T), hipMemcpyDeviceToDevice, st) != hipSuccess) r = ncclSystemError;
      }
      so += (size_t) scnt[p];
      ro += (size_t) rcnt[p];
   }
   ncclResult_t e = ncclGroupEnd ();
   if (r != ncclSuccess || e != ncclSuccess) c->failed = true;
   return (r != ncclSuccess || e != ncclSuccess);
}

int rccl_alltoallv (void *ctx, const void *dev_send, const int *scnt, void *dev_recv, const int *rcnt, void *stream)
{
   return exchange ((RcclCtx *) ctx, (const double *) dev_send, scnt, (double *) dev_recv, rcnt, ncclDouble, (hipStream_t) stream);
}

int rccl_alltoallv_i32_host (void *ctx, const int32_t *send, const int *scnt, int32_t *recv, const int *rcnt)
{
   RcclCtx *c = (RcclCtx *) ctx;
   size_t ns = 0, nr = 0;
   for (int p = 0; p < c->nranks; p++) { ns += (size_t) scnt[p]; nr += (size_t) rcnt[p]; }
   if (ensure_stage (c, (ns + nr + 2) * sizeof (int32_t))) return 1;
   int32_t *ds = (int32_t *) c->stage, *dr = ds + ns + 1;
   if (ns && hipMemcpy (ds, send, ns * sizeof (int32_t), hipMemcpyHostToDevice) != hipSuccess) return 1;
   if (exchange (c, (const int32_t *) ds, scnt, dr, rcnt, ncclInt32, (hipStream_t) 0)) return 1;
   if (hipStreamSynchronize (0) != hipSuccess) return 1;
   if (nr && hipMemcpy (recv, dr, nr * sizeof (int32_t), hipMemcpyDeviceToHost) != hipSuccess) return 1;
   return 0;
}

int rccl_allgather_i64_host (void *ctx, int64_t mine, int64_t *all)
{
   RcclCtx *c = (RcclCtx *) ctx;
   if (ensure_stage (c, ((size_t) c->nranks + 1) * sizeof (int64_t))) return 1;
   int64_t *d = (int64_t *) c->stage;
   if (hipMemcpy (d + c->nranks, &mine, sizeof mine, hipMemcpyHostToDevice) != hipSuccess) return 1;
   if (ncclAllGather (d + c->nranks, d, 1, ncclInt64, c->comm, (hipStream_t) 0) != ncclSuccess) { c->failed = true; return 1; }
   if (hipStreamSynchronize (0) != hipSuccess) return 1;
   return hipMemcpy (all, d, (size_t) c->nranks * sizeof (int64_t), hipMemcpyDeviceToHost) != hipSuccess;
}
}  // namespace

extern "C" int nkp_comm_unique_id (void *id128)
{
   static_assert (sizeof (ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
   if (!id128) return NKP_EINVAL;
   ncclUniqueId id;
   if (ncclGetUniqueId (&id) != ncclSuccess) return NKP_ECOMM;
   memcpy (id128, &id, sizeof id);
   return NKP_OK;
}

extern "C" int nkp_comm_rccl_init (nkp_comm_ops *ops, const void *id128, int rank, int nranks)
{
   if (!ops || !id128 || rank < 0 || rank >= nranks) return NKP_EINVAL;
   RcclCtx *c = new RcclCtx;
   c->rank = rank;
   c->nranks = nranks;
   ncclUniqueId id;
   memcpy (&id, id128, sizeof id);
   if (ncclCommInitRank (&c->comm, nranks, id, rank) != ncclSuccess) { delete c; return NKP_ECOMM; }
   ops->ctx = c;
   ops->rank = rank;
   ops->nranks = nranks;
   ops->allreduce = rccl_allreduce;
   ops->alltoallv = rccl_alltoallv;
   ops->alltoallv_i32_host = rccl_alltoallv_i32_host;
   ops->allgather_i64_host = rccl_allgather_i64_host;
   return NKP_OK;
}

extern "C" void nkp_comm_rccl_free (nkp_comm_ops *ops)
{
   if (!ops || !ops->ctx) return;
   RcclCtx *c = (RcclCtx *) ops->ctx;
   if (c->stage) (void) hipFree (c->stage);
   if (c->comm) {
      // ncclCommDestroy waits for the communicator's outstanding operations; after a failure (or an asynchronous error RCCL
      // has recorded) there may be one that never completes, and the peers must see this rank go away instead of waiting
      ncclResult_t async = ncclSuccess;
      if (ncclCommGetAsyncError (c->comm, &async) != ncclSuccess || async != ncclSuccess) c->failed = true;
      if (c->failed) (void) ncclCommAbort (c->comm);
      else (void) ncclCommDestroy (c->comm);
   }
   delete c;
   ops->ctx = nullptr;
}
