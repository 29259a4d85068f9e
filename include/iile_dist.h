/*
 * iile_dist.h — C ABI of libiile_dist.so: the one collective of the multi-GPU path.
 *
 * The reference fans SamplerIntegrator::Render's 16x16 tiles out over the threads of one process
 * (src/core/integrator.cpp:240-330, src/core/parallel.cpp:247-299) and merges every finished FilmTile into the
 * Film under a mutex (Film::MergeFilmTile, src/core/film.cpp:135-148: pixel += tile pixel, in XYZ + weight).
 * Here one process per GPU renders the tiles iile_tile_owner (iile_scene.h) gives it into a full-resolution
 * {X, Y, Z, filterWeightSum} film that is zero elsewhere (iile_render with tile_rank / tile_nranks), and
 * iile_dist_film_reduce stands where the mutex-protected merge stands: ONE sum-reduction of the films to the root
 * rank over RCCL (xGMI inside a node). Tiles are disjoint, so the sum adds values to zeros — except for the samples
 * whose film position is a whole number, which also land in a neighbouring pixel that may belong to another rank's
 * tile (the reference's one-pixel FilmTile halo, src/core/film.cpp:96-99).
 *
 * No torch types, no MPI: ranks find each other through a 128-byte RCCL unique id that rank 0 creates and the host
 * passes to the others by whatever channel it has (bench.py: a torch.distributed broadcast; iile_pbrt: a file, see
 * iile_dist_rendezvous_file). Every call returns 0 or an IILE_ERR_* code (iile_gpu.h), message in
 * iile_dist_last_error(). Thread-compatible, not thread-safe.
 */
#ifndef IILE_DIST_H
#define IILE_DIST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iile_dist iile_dist;

#define IILE_DIST_ID_BYTES 128

/* Rank 0: a fresh rendezvous id (ncclGetUniqueId). */
int iile_dist_unique_id(uint8_t id[IILE_DIST_ID_BYTES]);
/* Every rank, after hipSetDevice: join the communicator of `id` as `rank` of `nranks` (collective call). */
int iile_dist_create(const uint8_t id[IILE_DIST_ID_BYTES], int32_t rank, int32_t nranks, iile_dist **out);
/* The same for hosts with no launcher that would tear a stuck job down (the threads of one process, iile_pbrt --gpurank):
 * RCCL's non-blocking set-up (ncclCommInitRankConfig, blocking = 0) polled against a deadline. If a rank never arrives the call
 * returns IILE_ERR_TIMEOUT after timeout_s seconds — the reference's ParallelFor cannot lose a worker (src/core/parallel.cpp:
 * 247-299: threads of one pool); processes and devices can. The same deadline then bounds every wait on the communicator
 * (iile_dist_wait, _sum_u64, _max_f64, _all_ok, _rendezvous_done): when one expires the communicator is aborted locally
 * (ncclCommAbort) and every later call on it fails at once with IILE_ERR_TIMEOUT. */
int iile_dist_create_deadline(const uint8_t id[IILE_DIST_ID_BYTES], int32_t rank, int32_t nranks, double timeout_s, iile_dist **out);
void iile_dist_destroy(iile_dist *comm);
/* Leave without the peers' help (ncclCommAbort) and free the object: for a rank that knows the job is over. */
void iile_dist_abort(iile_dist *comm);
/* Everything queued on `stream` so far — the film merge — has completed. A communicator with a deadline waits that long at most
 * (then: aborted, IILE_ERR_TIMEOUT); one made by iile_dist_create waits as hipStreamSynchronize does. */
int iile_dist_wait(iile_dist *comm, void *stream);
int iile_dist_rank(const iile_dist *comm);
int iile_dist_size(const iile_dist *comm);
/* ncclCommCount of the communicator as RCCL reports it after ncclCommInitRank (0 if the query failed): bench.py prints it
 * beside the launcher's world size, so that a communicator that does not span every rank shows in the bench line. */
int iile_dist_ranks_seen(const iile_dist *comm);

/* The film merge: film_xyzw_dev (device memory, 4 floats per pixel, n_pixels pixels on every rank) is summed over
 * the ranks, in place, into rank `root`'s buffer; the other ranks' buffers are unchanged. Enqueued on `stream`
 * (a hipStream_t, NULL = the null stream) behind the render that filled the film; returns without waiting. */
int iile_dist_film_reduce(iile_dist *comm, float *film_xyzw_dev, int64_t n_pixels, int32_t root, void *stream);
/* The IISPT frame's film monitors (IisptFilmMonitor: sums of doubles per pixel, add_n_samples, iisptfilmmonitor.cpp:47-72) when the
 * frame's tasks and direct passes are shared out over the ranks: n_doubles doubles in device memory summed over the ranks, in place,
 * into rank `root`'s buffer; enqueued on `stream` like the film merge above. */
int iile_dist_monitor_reduce(iile_dist *comm, double *monitor_dev, int64_t n_doubles, int32_t root, void *stream);
/* All ranks have enqueued everything before it: completes on `stream` when every rank has reached it. */
int iile_dist_barrier(iile_dist *comm, void *stream);
/* Job totals for the host's report (ray counters, wall time): n values summed / maximised over the ranks in place,
 * on every rank; host memory in, host memory out (synchronous). */
int iile_dist_sum_u64(iile_dist *comm, uint64_t *values, int32_t n);
int iile_dist_max_f64(iile_dist *comm, double *values, int32_t n);

/* One status word agreed by all ranks (collective; synchronous): *all_ok = 1 iff every rank passed ok != 0. A rank that
 * failed locally (scene did not load, out of memory, render error) calls it with 0 instead of leaving the collective
 * sequence, so that the others do not wait for it inside the film merge. */
int iile_dist_all_ok(iile_dist *comm, int32_t ok, int32_t *all_ok);

/* Rendezvous for hosts without a launcher (iile_pbrt --gpurank r/n --rendezvous FILE [--job TOKEN]): rank 0 removes
 * whatever `path` holds, creates the id and publishes {magic, token, id} there (atomically: temporary file + rename);
 * the others wait (timeout_s seconds) for a file of THIS launch: with a token != 0 the file's token must match,
 * without one the file must not be older than the call (a file left behind by an earlier run is rejected either
 * way, never handed to ncclCommInitRank). iile_dist_rendezvous_file is the token-less form. */
int iile_dist_rendezvous_file_token(const char *path, int32_t rank, uint64_t token, uint8_t id[IILE_DIST_ID_BYTES], int32_t timeout_s);
int iile_dist_rendezvous_file(const char *path, int32_t rank, uint8_t id[IILE_DIST_ID_BYTES], int32_t timeout_s);
/* After iile_dist_create on every rank (collective): once all ranks have joined, rank 0 removes the file. */
int iile_dist_rendezvous_done(iile_dist *comm, const char *path);

const char *iile_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* IILE_DIST_H */
