// dist.cpp — libiile_dist.so (include/iile_dist.h): the film merge of the multi-GPU path over RCCL.
//
// One collective per frame: ncclReduce(sum) of the ranks' {X,Y,Z,w} films to the root, in place.
// xGMI is point to point (7 links per GPU), so a ring reduce of the 33 MB 1080p film is bound by one link:
// ~0.3 ms at 8 ranks — three orders of magnitude below the render; nothing here is worth overlapping or bucketing.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include <sys/stat.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../../include/iile_dist.h"
#include "../../../include/iile_gpu.h"  // status codes

namespace {
thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define NCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) return fail(IILE_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define HIPD_TRY(expr)                                                                                    \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(IILE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
static_assert(sizeof(ncclUniqueId) == IILE_DIST_ID_BYTES, "RCCL unique id size");
}  // namespace

struct iile_dist {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    // > 0: the communicator was made by iile_dist_create_deadline — non-blocking in RCCL's sense (a call may return
    // ncclInProgress; settle() polls it to its end), and nothing on it waits longer than this many seconds for the other ranks:
    // when a wait expires the communicator is aborted (ncclCommAbort: local, needs no peer), `dead` is set and every later call
    // fails at once. 0: made by iile_dist_create — RCCL's blocking calls; a launcher with its own tear-down owns the job.
    double timeout_s = 0.0;
    bool dead = false;
    void *scratch = nullptr;  // device staging of the small host-side totals
    size_t scratch_bytes = 0;
    // the host-value collectives (iile_dist_sum_u64 / _max_f64 / _all_ok) run on a non-blocking stream of the communicator's
    // own: the null stream they used until round 3 synchronises with every blocking stream of the process, i.e. with the render
    hipStream_t host_stream = nullptr;
    int ranks_seen = 0;       // ncclCommCount at creation: what RCCL itself thinks the communicator spans
};

namespace {
using Clock = std::chrono::steady_clock;
double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// give the communicator up without the peers' help; later calls on it fail at once
int expire(iile_dist *d, const std::string &what) {
    if (d->comm) (void)ncclCommAbort(d->comm);
    d->comm = nullptr;
    d->dead = true;
    return fail(IILE_ERR_TIMEOUT, what + ": no answer from the other ranks within " + std::to_string(int(d->timeout_s + 0.5)) +
                                      " s; the communicator was aborted (a rank that failed or never started?)");
}
// r = what an RCCL call on d->comm returned. Blocking communicators: success or an error. Non-blocking ones: ncclInProgress means
// "still being set up / enqueued" — poll ncclCommGetAsyncError until it settles or the communicator's deadline passes.
int settle(iile_dist *d, ncclResult_t r, const char *what) {
    if (r == ncclInProgress && d->timeout_s > 0) {
        const Clock::time_point t0 = Clock::now();
        for (;;) {
            ncclResult_t q = ncclInProgress;
            if (ncclCommGetAsyncError(d->comm, &q) != ncclSuccess) q = ncclSystemError;
            if (q != ncclInProgress) {
                r = q;
                break;
            }
            if (seconds_since(t0) > d->timeout_s) return expire(d, what);
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    if (r != ncclSuccess) return fail(IILE_ERR_HIP, std::string(what) + ": " + ncclGetErrorString(r));
    return IILE_OK;
}
// wait for `stream` (where a collective was queued): forever on a blocking communicator, at most the deadline on the other kind
int wait_stream(iile_dist *d, hipStream_t stream, const char *what) {
    if (d->timeout_s <= 0) {
        HIPD_TRY(hipStreamSynchronize(stream));
        return IILE_OK;
    }
    const Clock::time_point t0 = Clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return IILE_OK;
        if (e != hipErrorNotReady) return fail(IILE_ERR_HIP, std::string(what) + ": hipStreamQuery: " + hipGetErrorString(e));
        ncclResult_t q = ncclSuccess;
        if (ncclCommGetAsyncError(d->comm, &q) == ncclSuccess && q != ncclSuccess && q != ncclInProgress) {
            const std::string msg = std::string(what) + ": " + ncclGetErrorString(q);
            (void)ncclCommAbort(d->comm);
            d->comm = nullptr;
            d->dead = true;
            return fail(IILE_ERR_HIP, msg);
        }
        if (seconds_since(t0) > d->timeout_s) return expire(d, what);
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}
#define DEAD_CHECK(d, name) \
    if ((d)->dead || !(d)->comm) return fail(IILE_ERR_TIMEOUT, name ": the communicator was aborted by an earlier call")

int ensure_scratch(iile_dist *d, size_t bytes) {
    if (bytes <= d->scratch_bytes) return IILE_OK;
    if (d->scratch) HIPD_TRY(hipFree(d->scratch));
    d->scratch = nullptr;
    d->scratch_bytes = 0;
    HIPD_TRY(hipMalloc(&d->scratch, bytes));
    d->scratch_bytes = bytes;
    return IILE_OK;
}
template <typename T>
int all_reduce_host(iile_dist *d, T *values, int n, ncclDataType_t type, ncclRedOp_t op) {
    if (!d || !values || n < 0) return fail(IILE_ERR_ARG, "iile_dist: bad argument");
    DEAD_CHECK(d, "iile_dist");
    if (n == 0) return IILE_OK;
    int rc = ensure_scratch(d, size_t(n) * sizeof(T));
    if (rc) return rc;
    if (!d->host_stream) HIPD_TRY(hipStreamCreateWithFlags(&d->host_stream, hipStreamNonBlocking));
    HIPD_TRY(hipMemcpyAsync(d->scratch, values, size_t(n) * sizeof(T), hipMemcpyHostToDevice, d->host_stream));
    if ((rc = settle(d, ncclAllReduce(d->scratch, d->scratch, size_t(n), type, op, d->comm, d->host_stream), "ncclAllReduce"))) return rc;
    HIPD_TRY(hipMemcpyAsync(values, d->scratch, size_t(n) * sizeof(T), hipMemcpyDeviceToHost, d->host_stream));
    return wait_stream(d, d->host_stream, "iile_dist: host totals");  // the caller wants the values: it waits for THIS stream only
}
}  // namespace


extern "C" {

const char *iile_dist_last_error(void) { return g_err.c_str(); }

int iile_dist_unique_id(uint8_t id[IILE_DIST_ID_BYTES]) {
    if (!id) return fail(IILE_ERR_ARG, "iile_dist_unique_id: null argument");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return IILE_OK;
}

int iile_dist_create(const uint8_t id[IILE_DIST_ID_BYTES], int32_t rank, int32_t nranks, iile_dist **out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(IILE_ERR_ARG, "iile_dist_create: bad argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(IILE_ERR_NO_DEVICE, "iile_dist_create: no HIP device (the film merge runs on the GPUs)");
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    iile_dist *d = new iile_dist;
    d->rank = rank;
    d->size = nranks;
    ncclResult_t r = ncclCommInitRank(&d->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete d;
        return fail(IILE_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    if (ncclCommCount(d->comm, &d->ranks_seen) != ncclSuccess) d->ranks_seen = 0;
    *out = d;
    return IILE_OK;
}

// The communicator for hosts that have no launcher to tear a stuck job down (several threads of one process, iile_pbrt
// --gpurank): RCCL's non-blocking set-up (ncclCommInitRankConfig, blocking = 0) polled against a deadline. A rank that never
// arrives — it failed before this call, or was never started — makes every other rank's call return IILE_ERR_TIMEOUT after
// timeout_s seconds instead of waiting in ncclCommInitRank for ever, and the same deadline bounds every later wait on the
// communicator (iile_dist_wait, the host totals, iile_dist_all_ok).
int iile_dist_create_deadline(const uint8_t id[IILE_DIST_ID_BYTES], int32_t rank, int32_t nranks, double timeout_s, iile_dist **out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks || !(timeout_s > 0)) return fail(IILE_ERR_ARG, "iile_dist_create_deadline: bad argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(IILE_ERR_NO_DEVICE, "iile_dist_create: no HIP device (the film merge runs on the GPUs)");
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    iile_dist *d = new iile_dist;
    d->rank = rank;
    d->size = nranks;
    d->timeout_s = timeout_s;
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.blocking = 0;
    ncclResult_t r = ncclCommInitRankConfig(&d->comm, nranks, u, rank, &cfg);
    int rc = (r == ncclSuccess || r == ncclInProgress) && d->comm ? settle(d, r, "ncclCommInitRankConfig")
                                                                    : fail(IILE_ERR_HIP, std::string("ncclCommInitRankConfig: ") + ncclGetErrorString(r));
    if (rc) {
        const std::string keep = g_err;
        if (d->comm) (void)ncclCommAbort(d->comm);
        delete d;
        g_err = keep;
        return rc;
    }
    if (ncclCommCount(d->comm, &d->ranks_seen) != ncclSuccess) d->ranks_seen = 0;
    *out = d;
    return IILE_OK;
}

void iile_dist_destroy(iile_dist *d) {
    if (!d) return;
    if (d->host_stream) (void)hipStreamDestroy(d->host_stream);
    if (d->scratch) (void)hipFree(d->scratch);
    if (d->comm) {
        // a non-blocking communicator is finalised first and given its deadline to finish; one that does not is aborted
        if (d->timeout_s > 0) {
            if (settle(d, ncclCommFinalize(d->comm), "ncclCommFinalize") == IILE_OK && d->comm) (void)ncclCommDestroy(d->comm);
            else if (d->comm) (void)ncclCommAbort(d->comm);
        } else {
            (void)ncclCommDestroy(d->comm);
        }
    }
    delete d;
}

// Leave without the peers: ncclCommAbort frees the communicator locally whatever the other ranks are doing (a rank that
// learned — from its own error or from the gang's vote — that the job is over must not wait in ncclCommDestroy for them).
void iile_dist_abort(iile_dist *d) {
    if (!d) return;
    if (d->comm) (void)ncclCommAbort(d->comm);
    d->comm = nullptr;
    iile_dist_destroy(d);
}

// Everything queued on `stream` so far (the film merge) has completed — or the communicator's deadline has passed, in which
// case it is aborted and the call fails. On a communicator made by iile_dist_create this is hipStreamSynchronize.
int iile_dist_wait(iile_dist *d, void *stream) {
    if (!d) return fail(IILE_ERR_ARG, "iile_dist_wait: null communicator");
    DEAD_CHECK(d, "iile_dist_wait");
    return wait_stream(d, static_cast<hipStream_t>(stream), "iile_dist_wait (film merge)");
}

int iile_dist_rank(const iile_dist *d) { return d ? d->rank : 0; }
int iile_dist_size(const iile_dist *d) { return d ? d->size : 1; }
int iile_dist_ranks_seen(const iile_dist *d) { return d ? d->ranks_seen : 0; }

int iile_dist_film_reduce(iile_dist *d, float *film, int64_t n_pixels, int32_t root, void *stream) {
    if (!d || !film || n_pixels < 0 || root < 0 || root >= d->size) return fail(IILE_ERR_ARG, "iile_dist_film_reduce: bad argument");
    DEAD_CHECK(d, "iile_dist_film_reduce");
    if (n_pixels == 0) return IILE_OK;
    // in place: sendbuff == recvbuff; RCCL leaves the non-root buffers as they are
    return settle(d, ncclReduce(film, film, size_t(n_pixels) * 4, ncclFloat32, ncclSum, root, d->comm, static_cast<hipStream_t>(stream)), "ncclReduce");
}

int iile_dist_monitor_reduce(iile_dist *d, double *monitor, int64_t n_doubles, int32_t root, void *stream) {
    if (!d || !monitor || n_doubles < 0 || root < 0 || root >= d->size) return fail(IILE_ERR_ARG, "iile_dist_monitor_reduce: bad argument");
    DEAD_CHECK(d, "iile_dist_monitor_reduce");
    if (n_doubles == 0) return IILE_OK;
    return settle(d, ncclReduce(monitor, monitor, size_t(n_doubles), ncclFloat64, ncclSum, root, d->comm, static_cast<hipStream_t>(stream)), "ncclReduce");
}

int iile_dist_barrier(iile_dist *d, void *stream) {
    if (!d) return fail(IILE_ERR_ARG, "iile_dist_barrier: null communicator");
    DEAD_CHECK(d, "iile_dist_barrier");
    int rc = ensure_scratch(d, 256);
    if (rc) return rc;
    return settle(d, ncclAllReduce(d->scratch, d->scratch, 1, ncclInt32, ncclSum, d->comm, static_cast<hipStream_t>(stream)), "ncclAllReduce");
}

int iile_dist_sum_u64(iile_dist *d, uint64_t *values, int32_t n) { return all_reduce_host(d, values, n, ncclUint64, ncclSum); }
int iile_dist_max_f64(iile_dist *d, double *values, int32_t n) { return all_reduce_host(d, values, n, ncclFloat64, ncclMax); }

// File layout: "IILEDIST" (8 bytes), the job token (u64, 0 = none), the RCCL unique id (128 bytes).
namespace {
constexpr char kRvMagic[8] = {'I', 'I', 'L', 'E', 'D', 'I', 'S', 'T'};
constexpr size_t kRvBytes = 16 + IILE_DIST_ID_BYTES;
}  // namespace

namespace {
// when this library was loaded: for a program linked against it (iile_pbrt) that is process start, before any HIP initialisation
const std::chrono::system_clock::time_point g_loaded_at = std::chrono::system_clock::now();
constexpr int kRvSlackSeconds = 5;  // file-system time stamps, small clock differences between launcher and ranks
}  // namespace

int iile_dist_rendezvous_file_token(const char *path, int32_t rank, uint64_t token, uint8_t id[IILE_DIST_ID_BYTES], int32_t timeout_s) {
    if (!path || !id || rank < 0) return fail(IILE_ERR_ARG, "iile_dist_rendezvous_file: bad argument");
    if (rank == 0) {
        // a file left by an earlier run holds a dead id: it goes before anything of this run can be read
        (void)std::remove(path);
        int rc = iile_dist_unique_id(id);
        if (rc) return rc;
        const std::string tmp = std::string(path) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f) return fail(IILE_ERR_ARG, "iile_dist_rendezvous_file: cannot write " + tmp);
        bool ok = std::fwrite(kRvMagic, 1, 8, f) == 8 && std::fwrite(&token, 1, 8, f) == 8;
        ok = ok && std::fwrite(id, 1, IILE_DIST_ID_BYTES, f) == IILE_DIST_ID_BYTES;
        if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path) != 0)
            return fail(IILE_ERR_ARG, std::string("iile_dist_rendezvous_file: cannot publish ") + path);
        return IILE_OK;
    }
    // Ranks != 0 accept a file only if it belongs to THIS launch (a rank that starts before rank 0 has removed a previous
    // run's file would otherwise pick up that run's dead id and hang in ncclCommInitRank). With a token: the token must match,
    // any start order and any delay work. Without one: (a) a file that was not there, or was a different file, when this call
    // began was written since — rank 0 of this launch published it —: accepted whenever it appears; (b) a file that WAS there
    // is either a previous run's or this launch's rank 0 having been quicker: accepted if it is not older than this process
    // (the library's load time, not this call's: HIP initialisation between the two can take seconds when eight ranks start
    // together, which the round-3 rule — the reader's clock at call time, one second of slack — turned into a 120 s timeout).
    // What stays out of reach without a token: a rank started more than kRvSlackSeconds after rank 0 has published (by hand,
    // over ssh). The error message says so; tools/multi_gpu_cmdline.sh always passes --job.
    const auto t0 = std::chrono::steady_clock::now();
    std::string why = "no file";
    struct stat at_entry;
    const bool had_file = stat(path, &at_entry) == 0;
    for (;;) {
        struct stat sb;
        if (FILE *f = std::fopen(path, "rb")) {
            unsigned char buf[kRvBytes];
            const size_t n = std::fread(buf, 1, kRvBytes, f);
            const bool have_stat = fstat(fileno(f), &sb) == 0;
            std::fclose(f);
            uint64_t file_token = 0;
            if (n == kRvBytes && std::memcmp(buf, kRvMagic, 8) == 0) {
                std::memcpy(&file_token, buf + 8, 8);
                bool fresh;
                if (token != 0) {
                    fresh = file_token == token;
                    if (!fresh) why = "a file of another launch (token mismatch)";
                } else {
                    const auto mtime = std::chrono::system_clock::from_time_t(have_stat ? sb.st_mtime : 0);
                    const bool replaced = have_stat && (!had_file || sb.st_ino != at_entry.st_ino || sb.st_mtim.tv_sec != at_entry.st_mtim.tv_sec ||
                                                        sb.st_mtim.tv_nsec != at_entry.st_mtim.tv_nsec);
                    fresh = have_stat && file_token == 0 && (replaced || mtime + std::chrono::seconds(kRvSlackSeconds) >= g_loaded_at);
                    if (!fresh)
                        why = file_token != 0 ? "a file published with a token"
                                              : "a file older than this process (a previous run's; or this rank started long after rank 0: pass the "
                                                "same --job token to every rank)";
                }
                if (fresh) {
                    std::memcpy(id, buf + 16, IILE_DIST_ID_BYTES);
                    return IILE_OK;
                }
            } else {
                why = "a file that is not a rendezvous record";
            }
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > double(timeout_s))
            return fail(IILE_ERR_ARG, std::string("iile_dist_rendezvous_file: timed out waiting for ") + path + " (last seen: " + why + ")");
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}

int iile_dist_rendezvous_file(const char *path, int32_t rank, uint8_t id[IILE_DIST_ID_BYTES], int32_t timeout_s) {
    return iile_dist_rendezvous_file_token(path, rank, 0, id, timeout_s);
}

int iile_dist_rendezvous_done(iile_dist *d, const char *path) {
    if (!d || !path) return fail(IILE_ERR_ARG, "iile_dist_rendezvous_done: bad argument");
    // every rank has joined once a collective over the communicator completes; then nobody reads the file any more
    uint64_t one = 1;
    int rc = iile_dist_sum_u64(d, &one, 1);
    if (rc) return rc;
    if (one != uint64_t(d->size)) return fail(IILE_ERR_HIP, "iile_dist_rendezvous_done: the communicator does not span every rank");
    if (d->rank == 0) (void)std::remove(path);
    return IILE_OK;
}

int iile_dist_all_ok(iile_dist *d, int32_t ok, int32_t *all_ok) {
    if (!d || !all_ok) return fail(IILE_ERR_ARG, "iile_dist_all_ok: bad argument");
    uint64_t failed = ok ? 0 : 1;
    int rc = iile_dist_sum_u64(d, &failed, 1);
    if (rc) return rc;
    *all_ok = failed == 0 ? 1 : 0;
    return IILE_OK;
}

}  // extern "C"
