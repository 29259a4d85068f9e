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
void iile_dist_destroy(iile_dist *comm);
int iile_dist_rank(const iile_dist *comm);
int iile_dist_size(const iile_dist *comm);

/* The film merge: film_xyzw_dev (device memory, 4 floats per pixel, n_pixels pixels on every rank) is summed over
 * the ranks, in place, into rank `root`'s buffer; the other ranks' buffers are unchanged. Enqueued on `stream`
 * (a hipStream_t, NULL = the null stream) behind the render that filled the film; returns without waiting. */
int iile_dist_film_reduce(iile_dist *comm, float *film_xyzw_dev, int64_t n_pixels, int32_t root, void *stream);
/* All ranks have enqueued everything before it: completes on `stream` when every rank has reached it. */
int iile_dist_barrier(iile_dist *comm, void *stream);
/* Job totals for the host's report (ray counters, wall time): n values summed / maximised over the ranks in place,
 * on every rank; host memory in, host memory out (synchronous). */
int iile_dist_sum_u64(iile_dist *comm, uint64_t *values, int32_t n);
int iile_dist_max_f64(iile_dist *comm, double *values, int32_t n);

/* Rendezvous for hosts without a launcher (iile_pbrt --gpurank r/n --rendezvous FILE): rank 0 creates the id and
 * writes it to `path` (atomically: temporary file + rename), the others wait until it appears (timeout_s seconds). */
int iile_dist_rendezvous_file(const char *path, int32_t rank, uint8_t id[IILE_DIST_ID_BYTES], int32_t timeout_s);

const char *iile_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* IILE_DIST_H */
