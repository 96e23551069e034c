// Host-staged implementation of nkp_comm_ops (include/nkp.h) over a shared directory: every collective is one file per
// rank and sequence number.  It exists so that the multi-process code path of bin/solve_ABdist (partition, halo plan,
// distributed Krylov iteration, gather on rank 0 -- reference src/solve_ABdist.c:115-244, 334-418) can be exercised with
// several processes on a box that has ONE GPU, where RCCL cannot form a communicator.  Correct, slow, test-grade: the
// production transport is comm_rccl.hip.
//
// Protocol: collective number q on rank r writes <dir>/q.r (atomically: tmp + rename) = [header | its whole send
// buffer], then reads <dir>/q.p of every peer p.  A rank deletes its file of collective q - 2 once it has read every
// peer's file of q - 1 (a peer that has written q - 1 is done reading q - 2).  Reads poll with a deadline
// (NKP_COMM_TIMEOUT seconds, default 120): a peer that died makes the collective fail instead of hanging the job.
#include "../../include/nkp.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace {
struct FileCtx {
   std::string dir;
   int rank = 0, nranks = 1;
   long seq = 0;
   double timeout = 120.0;
};

struct Header { long seq; int rank, nranks; long bytes; };

std::string path_of (const FileCtx *c, long seq, int rank) { return c->dir + "/" + std::to_string (seq) + "." + std::to_string (rank); }

double now () { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec; }

int publish (FileCtx *c, const void *buf, size_t bytes)
{
   const std::string fin = path_of (c, c->seq, c->rank), tmp = fin + ".tmp";
   FILE *f = fopen (tmp.c_str (), "wb");
   if (!f) return 1;
   Header h = { c->seq, c->rank, c->nranks, (long) bytes };
   const bool ok = fwrite (&h, sizeof h, 1, f) == 1 && (bytes == 0 || fwrite (buf, 1, bytes, f) == bytes);
   if (fclose (f) || !ok) return 1;
   return rename (tmp.c_str (), fin.c_str ()) != 0;
}

// whole payload of peer p for the current collective
int fetch (FileCtx *c, int p, std::vector<char> &out)
{
   const std::string fn = path_of (c, c->seq, p);
   const double deadline = now () + c->timeout;
   for (;;) {
      FILE *f = fopen (fn.c_str (), "rb");
      if (f) {
         Header h;
         bool ok = fread (&h, sizeof h, 1, f) == 1 && h.seq == c->seq && h.rank == p && h.nranks == c->nranks && h.bytes >= 0;
         if (ok) {
            out.resize ((size_t) h.bytes);
            ok = h.bytes == 0 || fread (out.data (), 1, (size_t) h.bytes, f) == (size_t) h.bytes;
         }
         fclose (f);
         return ok ? 0 : 1;
      }
      if (now () > deadline) return 1;
      usleep (200);
   }
}

void retire (FileCtx *c)
{
   if (c->seq >= 2) (void) unlink (path_of (c, c->seq - 2, c->rank).c_str ());
   c->seq++;
}

// personalised exchange of raw bytes between host buffers (elem = bytes per item)
int exchange_host (FileCtx *c, const char *send, const int *scnt, char *recv, const int *rcnt, size_t elem)
{
   size_t ns = 0;
   for (int p = 0; p < c->nranks; p++) ns += (size_t) scnt[p];
   // payload = the send counts, then the whole send buffer: every peer cuts out its own piece
   std::vector<char> mine (sizeof (int) * (size_t) c->nranks + ns * elem);
   memcpy (mine.data (), scnt, sizeof (int) * (size_t) c->nranks);
   if (ns) memcpy (mine.data () + sizeof (int) * (size_t) c->nranks, send, ns * elem);
   if (publish (c, mine.data (), mine.size ())) return 1;
   size_t ro = 0;
   std::vector<char> theirs;
   for (int p = 0; p < c->nranks; p++) {
      const std::vector<char> *src = &mine;
      if (p != c->rank) {
         if (fetch (c, p, theirs)) return 1;
         src = &theirs;
      }
      const int *pc = (const int *) src->data ();
      size_t off = 0;
      for (int q = 0; q < c->rank; q++) off += (size_t) pc[q];
      if (pc[c->rank] != rcnt[p]) return 1;                       // the two sides disagree about the plan
      if (rcnt[p]) memcpy (recv + ro * elem, src->data () + sizeof (int) * (size_t) c->nranks + off * elem, (size_t) rcnt[p] * elem);
      ro += (size_t) rcnt[p];
   }
   retire (c);
   return 0;
}

int file_allreduce (void *ctx, void *dev_buf, int count, int op, void *stream)
{
   FileCtx *c = (FileCtx *) ctx;
   std::vector<double> mine ((size_t) count), acc ((size_t) count);
   if (hipMemcpyAsync (mine.data (), dev_buf, (size_t) count * sizeof (double), hipMemcpyDeviceToHost, (hipStream_t) stream) != hipSuccess) return 1;
   if (hipStreamSynchronize ((hipStream_t) stream) != hipSuccess) return 1;
   if (publish (c, mine.data (), mine.size () * sizeof (double))) return 1;
   std::vector<char> theirs;
   // fixed rank order => every rank computes the same bits
   for (int p = 0; p < c->nranks; p++) {
      const double *v = mine.data ();
      if (p != c->rank) {
         if (fetch (c, p, theirs) || theirs.size () != (size_t) count * sizeof (double)) return 1;
         v = (const double *) theirs.data ();
      }
      for (int i = 0; i < count; i++) acc[(size_t) i] = p == 0 ? v[i] : (op == 1 ? (v[i] > acc[(size_t) i] ? v[i] : acc[(size_t) i]) : acc[(size_t) i] + v[i]);
   }
   retire (c);
   if (hipMemcpyAsync (dev_buf, acc.data (), (size_t) count * sizeof (double), hipMemcpyHostToDevice, (hipStream_t) stream) != hipSuccess) return 1;
   return hipStreamSynchronize ((hipStream_t) stream) != hipSuccess;
}

int file_alltoallv (void *ctx, const void *dev_send, const int *scnt, void *dev_recv, const int *rcnt, void *stream)
{
   FileCtx *c = (FileCtx *) ctx;
   size_t ns = 0, nr = 0;
   for (int p = 0; p < c->nranks; p++) { ns += (size_t) scnt[p]; nr += (size_t) rcnt[p]; }
   std::vector<double> hs (ns + 1), hr (nr + 1);
   if (ns && hipMemcpyAsync (hs.data (), dev_send, ns * sizeof (double), hipMemcpyDeviceToHost, (hipStream_t) stream) != hipSuccess) return 1;
   if (hipStreamSynchronize ((hipStream_t) stream) != hipSuccess) return 1;
   if (exchange_host (c, (const char *) hs.data (), scnt, (char *) hr.data (), rcnt, sizeof (double))) return 1;
   if (nr && hipMemcpyAsync (dev_recv, hr.data (), nr * sizeof (double), hipMemcpyHostToDevice, (hipStream_t) stream) != hipSuccess) return 1;
   return hipStreamSynchronize ((hipStream_t) stream) != hipSuccess;
}

int file_alltoallv_i32_host (void *ctx, const int32_t *send, const int *scnt, int32_t *recv, const int *rcnt)
{
   return exchange_host ((FileCtx *) ctx, (const char *) send, scnt, (char *) recv, rcnt, sizeof (int32_t));
}

int file_allgather_i64_host (void *ctx, int64_t mine, int64_t *all)
{
   FileCtx *c = (FileCtx *) ctx;
   if (publish (c, &mine, sizeof mine)) return 1;
   std::vector<char> theirs;
   for (int p = 0; p < c->nranks; p++) {
      if (p == c->rank) { all[p] = mine; continue; }
      if (fetch (c, p, theirs) || theirs.size () != sizeof (int64_t)) return 1;
      memcpy (&all[p], theirs.data (), sizeof (int64_t));
   }
   retire (c);
   return 0;
}
}  // namespace

extern "C" int nkp_comm_file_init (nkp_comm_ops *ops, const char *dir, int rank, int nranks)
{
   if (!ops || !dir || !*dir || rank < 0 || rank >= nranks) return NKP_EINVAL;
   if (access (dir, W_OK) != 0) return NKP_ECOMM;
   FileCtx *c = new FileCtx;
   c->dir = dir;
   c->rank = rank;
   c->nranks = nranks;
   if (const char *e = getenv ("NKP_COMM_TIMEOUT")) { const double t = atof (e); if (t > 0.0) c->timeout = t; }
   ops->ctx = c;
   ops->rank = rank;
   ops->nranks = nranks;
   ops->allreduce = file_allreduce;
   ops->alltoallv = file_alltoallv;
   ops->alltoallv_i32_host = file_alltoallv_i32_host;
   ops->allgather_i64_host = file_allgather_i64_host;
   return NKP_OK;
}

extern "C" void nkp_comm_file_free (nkp_comm_ops *ops)
{
   if (!ops || !ops->ctx) return;
   FileCtx *c = (FileCtx *) ops->ctx;
   // Teardown without a race: a rank that has left its last collective may be far ahead of a peer that is still reading
   // this rank's file of that collective (nkp_gather_root: rank 0 fetches everybody's slice last).  So every rank first
   // takes part in one more allgather -- a peer that has WRITTEN its file of it has finished reading everything before --
   // and removes its older files; the files of that last allgather stay until every rank > 0 has dropped a "done" token
   // (written after its last read), then rank 0 removes them all.  A peer that died shows as a missed deadline: its files
   // are left behind rather than waited for.
   std::vector<int64_t> all ((size_t) c->nranks + 1, 0);
   const long last = c->seq;
   const bool joined = file_allgather_i64_host (c, 0, all.data ()) == 0;
   for (long q = last >= 2 ? last - 2 : 0; q < last; q++) (void) unlink (path_of (c, q, c->rank).c_str ());
   if (joined) {
      const std::string done = c->dir + "/done.";
      if (c->rank != 0) {
         FILE *f = fopen ((done + std::to_string (c->rank)).c_str (), "wb");
         if (f) fclose (f);
      } else {
         const double deadline = now () + c->timeout;
         bool everyone = true;
         for (int p = 1; p < c->nranks && everyone; p++) {
            const std::string fn = done + std::to_string (p);
            while (access (fn.c_str (), F_OK) != 0) {
               if (now () > deadline) { everyone = false; break; }
               usleep (200);
            }
         }
         if (everyone)
            for (int p = 0; p < c->nranks; p++) {
               (void) unlink (path_of (c, last, p).c_str ());
               if (p) (void) unlink ((done + std::to_string (p)).c_str ());
            }
      }
   }
   delete c;
   ops->ctx = nullptr;
}
