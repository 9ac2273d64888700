// api_comm.hip -- C ABI, part 6: the RCCL exchange steps (vocabulary broadcast, all-gather + merge of sharded brute force).
#include "api_common.h"

// ------------------------------------------------------------------------------------------------
// RCCL (loaded lazily so that the library has no hard link-time dependency on it)
// ------------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } rccl_uid_t;
typedef int (*fn_getuid)(rccl_uid_t *);
typedef int (*fn_initrank)(void **, int, rccl_uid_t, int);
typedef int (*fn_bcast)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);
typedef int (*fn_commint)(void *, int *);
static struct {
    void *h = nullptr;
    bool tried = false, ok = false;
    fn_getuid getuid = nullptr;
    fn_initrank initrank = nullptr;
    fn_bcast bcast = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
    fn_commint count = nullptr, userrank = nullptr;   // optional (orbhip_comm_info)
} g_rccl;
static std::mutex g_rccl_mutex;

// One attempt per process; the outcome (every required symbol resolved) is what later calls see.
static bool rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.tried) return g_rccl.ok;
    g_rccl.tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) return false;
    g_rccl.getuid = (fn_getuid)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.initrank = (fn_initrank)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.bcast = (fn_bcast)dlsym(g_rccl.h, "ncclBroadcast");
    g_rccl.allgather = (fn_allgather)dlsym(g_rccl.h, "ncclAllGather");
    g_rccl.destroy = (fn_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(g_rccl.h, "ncclGetErrorString");
    g_rccl.count = (fn_commint)dlsym(g_rccl.h, "ncclCommCount");
    g_rccl.userrank = (fn_commint)dlsym(g_rccl.h, "ncclCommUserRank");
    g_rccl.ok = g_rccl.getuid && g_rccl.initrank && g_rccl.bcast && g_rccl.allgather && g_rccl.destroy;
    return g_rccl.ok;
}

static std::string rccl_err(const char *what, int rc)
{
    return std::string(what) + ": " + (g_rccl.errstr ? g_rccl.errstr(rc) : "error");
}

// called by orbhip_destroy (above): the communicator belongs to the context
void orb_comm_release(orbhip_ctx *c)
{
    if (c->comm && g_rccl.ok) (void)g_rccl.destroy(c->comm);
    c->comm = nullptr;
    c->nranks = 1;
    c->rank = 0;
}

extern "C" int orbhip_comm_unique_id(uint8_t uid[128])
{
    if (!uid) return ORBHIP_E_ARG;
    if (!rccl_load()) return fail(nullptr, ORBHIP_E_COMM, "cannot load librccl (or it lacks a required symbol)");
    rccl_uid_t u;
    int rc = g_rccl.getuid(&u);
    if (rc != 0) return fail(nullptr, ORBHIP_E_COMM, rccl_err("ncclGetUniqueId", rc));
    memcpy(uid, u.internal, 128);
    return ORBHIP_OK;
}

extern "C" int orbhip_comm_init(orbhip_ctx *c, int rank, int nranks, const uint8_t uid[128])
{
    if (!c || !uid || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, ORBHIP_E_ARG, "orbhip_comm_init: bad argument");
    if (!rccl_load()) return fail(c, ORBHIP_E_COMM, "cannot load librccl (or it lacks a required symbol)");
    HIPCHK(c, orb_enter(c));
    orb_comm_release(c);
    rccl_uid_t u;
    memcpy(u.internal, uid, 128);
    int rc = g_rccl.initrank(&c->comm, nranks, u, rank);
    if (rc != 0) {
        c->comm = nullptr;
        return fail(c, ORBHIP_E_COMM, rccl_err("ncclCommInitRank", rc));
    }
    c->rank = rank;
    c->nranks = nranks;
    return ORBHIP_OK;
}

extern "C" int orbhip_comm_destroy(orbhip_ctx *c)
{
    if (!c) return ORBHIP_E_ARG;
    HIPCHK(c, orb_enter(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    orb_comm_release(c);
    return ORBHIP_OK;
}

extern "C" int orbhip_comm_info(orbhip_ctx *c, int *rank, int *nranks)
{
    if (!c) return ORBHIP_E_ARG;
    int r = c->rank, n = c->nranks;
    if (c->comm && g_rccl.count && g_rccl.userrank) {
        // what the communicator itself reports (ncclCommCount / ncclCommUserRank), not what the caller passed in
        int rc = g_rccl.count(c->comm, &n);
        if (rc == 0) rc = g_rccl.userrank(c->comm, &r);
        if (rc != 0) return fail(c, ORBHIP_E_COMM, rccl_err("ncclCommCount / ncclCommUserRank", rc));
    }
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    return ORBHIP_OK;
}

extern "C" int orbhip_bcast_blob_device(orbhip_ctx *c, void *d_buf, size_t nbytes, int root)
{
    if (!c || !d_buf) return fail(c, ORBHIP_E_ARG, "orbhip_bcast_blob_device: bad argument");
    if (!c->comm) {
        if (c->nranks == 1) return ORBHIP_OK;   // no communicator and a single rank: nothing to exchange
        return fail(c, ORBHIP_E_COMM, "orbhip_comm_init was not called");
    }
    if (root < 0 || root >= c->nranks) return fail(c, ORBHIP_E_ARG, "orbhip_bcast_blob_device: bad root");
    HIPCHK(c, orb_enter(c));
    // ncclChar = 0; with a communicator the collective runs also for one rank (an in-place no-op that exercises the RCCL path)
    int rc = g_rccl.bcast(d_buf, d_buf, nbytes, 0, root, c->comm, c->stream);
    if (rc != 0) return fail(c, ORBHIP_E_COMM, rccl_err("ncclBroadcast", rc));
    return ORBHIP_OK;
}

extern "C" int orbhip_knn2_merge_device(orbhip_ctx *c, const void *d_parts, int nshards, int nq, void *d_best_idx,
                                        void *d_best_d, void *d_second_d)
{
    if (!c || nshards < 1 || nq < 0 || (nq > 0 && (!d_parts || !d_best_idx || !d_best_d || !d_second_d)))
        return fail(c, ORBHIP_E_ARG, "orbhip_knn2_merge_device: bad argument");
    if (nq == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    launch_knn2_merge(c->stream, (const int32_t *)d_parts, nshards, nq, (int32_t *)d_best_idx, (int32_t *)d_best_d,
                      (int32_t *)d_second_d);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_knn2_allgather_merge_device(orbhip_ctx *c, const void *d_best_idx_local, const void *d_best_d_local,
                                                  const void *d_second_d_local, int nq, int shard_offset, void *d_best_idx,
                                                  void *d_best_d, void *d_second_d)
{
    if (!c || nq < 0 || (nq > 0 && (!d_best_idx_local || !d_best_d_local || !d_second_d_local || !d_best_idx || !d_best_d ||
                                    !d_second_d)))
        return fail(c, ORBHIP_E_ARG, "orbhip_knn2_allgather_merge_device: bad argument");
    if (nq == 0) return ORBHIP_OK;
    if (c->nranks > 1 && !c->comm) return fail(c, ORBHIP_E_COMM, "orbhip_comm_init was not called");
    HIPCHK(c, orb_enter(c));
    // scratch: my part [3 * nq + 1] then the gathered parts [nranks][3 * nq + 1]; part = best_idx | best_d | second_d | offset
    const size_t part = (size_t)3 * nq + 1;
    int rc;
    if ((rc = orb_match_scratch(c, (part * (size_t)(c->nranks + 1)) * 4 + 256))) return rc;
    int32_t *mine = (int32_t *)c->d_match, *all = mine + part;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(mine, d_best_idx_local, (size_t)nq * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(mine + nq, d_best_d_local, (size_t)nq * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(mine + 2 * (size_t)nq, d_second_d_local, (size_t)nq * 4, hipMemcpyDeviceToDevice, s));
    launch_fill_i32(s, mine + 3 * (size_t)nq, shard_offset, 1);
    if (c->comm) {
        // ncclInt32 = 2: the one exchange step of database-sharded brute force, Q x 12 bytes per rank
        int nrc = g_rccl.allgather(mine, all, part, 2, c->comm, s);
        if (nrc != 0) return fail(c, ORBHIP_E_COMM, rccl_err("ncclAllGather", nrc));
    } else {
        HIPCHK(c, hipMemcpyAsync(all, mine, part * 4, hipMemcpyDeviceToDevice, s));
    }
    launch_knn2_merge(s, all, c->nranks, nq, (int32_t *)d_best_idx, (int32_t *)d_best_d, (int32_t *)d_second_d);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}
