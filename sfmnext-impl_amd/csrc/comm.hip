// comm.hip — the gradient exchange of one-process-per-GPU training: a RCCL communicator behind the C ABI.
//
// Replaces nn.DataParallel's scatter / gather of the reference (trainer.py:73-74,92-93).  The
// collectives are enqueued on the CALLER's stream, so they order, overlap and capture into a hipGraph exactly like the
// kernels of this library; there is no helper thread polling the caller's events (torch's ProcessGroupNCCL watchdog
// aborted round 2's in-capture exchange on a fresh MI355X: hipErrorCapturedEvent raised from an event query).
//
// librccl is bound with dlopen/dlsym at run time: a PyTorch process passes torch/lib/librccl.so (sqd_comm_load), which is
// already mapped and bound to torch's libamdhip64 — a second RCCL or a second HIP runtime in the process would own
// different device contexts.  libsqd.so itself has no link-time dependency on RCCL.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "sqd_common.h"

namespace {

// the slice of rccl.h (NCCL 2.27 API) this file uses; the enums' numeric values are fixed by that API
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[SQD_COMM_ID_BYTES];
} ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1, ncclInt32 = 2, ncclFloat32 = 7, ncclFloat64 = 8 };
enum { ncclSum = 0, ncclMax = 2, ncclMin = 3, ncclAvg = 4 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    const char *(*GetLastError)(ncclComm_t) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int bind_rccl(const char *path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return SQD_OK;
    const char *name = (path && path[0]) ? path : "librccl.so";
    void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h && !(path && path[0])) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        sqd::set_error("sqd_comm_load: dlopen(%s) failed: %s", name, dlerror());
        return SQD_ECOMM;
    }
    Rccl r;
    r.handle = h;
#define SQD_BIND(field, sym)                                                          \
    do {                                                                              \
        *(void **)(&r.field) = dlsym(h, sym);                                         \
        if (!r.field) {                                                               \
            sqd::set_error("sqd_comm_load: %s has no symbol %s", name, sym);          \
            dlclose(h);                                                               \
            return SQD_ECOMM;                                                         \
        }                                                                             \
    } while (0)
    SQD_BIND(GetUniqueId, "ncclGetUniqueId");
    SQD_BIND(CommInitRank, "ncclCommInitRank");
    SQD_BIND(CommDestroy, "ncclCommDestroy");
    SQD_BIND(AllReduce, "ncclAllReduce");
    SQD_BIND(Broadcast, "ncclBroadcast");
    SQD_BIND(GetErrorString, "ncclGetErrorString");
#undef SQD_BIND
    *(void **)(&r.GetLastError) = dlsym(h, "ncclGetLastError");      // optional (absent before NCCL 2.13)
    *(void **)(&r.GetVersion) = dlsym(h, "ncclGetVersion");          // optional: only reported (sqd_comm_rccl_version)
    *(void **)(&r.CommCount) = dlsym(h, "ncclCommCount");            // optional: only reported (sqd_comm_joined)
    g_rccl = r;
    return SQD_OK;
}

int rccl_dtype(int dtype, size_t *size) {
    switch (dtype) {
        case 0: *size = 4; return ncclFloat32;
        case 1: *size = 8; return ncclFloat64;
        case 2: *size = 4; return ncclInt32;
        case 3: *size = 1; return ncclUint8;
    }
    return -1;
}

}  // namespace

struct sqd_comm {
    ncclComm_t nccl;
    int rank, world;
};

#define SQD_CHECK_RCCL(call, what, comm)                                                                        \
    do {                                                                                                        \
        int rc_ = (call);                                                                                       \
        if (rc_ != ncclSuccess) {                                                                               \
            const char *detail_ = g_rccl.GetLastError ? g_rccl.GetLastError(comm) : "";                         \
            sqd::set_error("%s: RCCL error %d (%s) %s", what, rc_, g_rccl.GetErrorString(rc_), detail_ ? detail_ : ""); \
            return SQD_ECOMM;                                                                                   \
        }                                                                                                       \
    } while (0)

extern "C" int sqd_comm_load(const char *librccl_path_host) { return bind_rccl(librccl_path_host); }

extern "C" int sqd_comm_unique_id(void *id_host) {
    SQD_CHECK_ARG(id_host, "sqd_comm_unique_id: null id buffer");
    if (int rc = bind_rccl(nullptr)) return rc;
    ncclUniqueId id;
    SQD_CHECK_RCCL(g_rccl.GetUniqueId(&id), "sqd_comm_unique_id", nullptr);
    memcpy(id_host, id.internal, SQD_COMM_ID_BYTES);
    return SQD_OK;
}

extern "C" int sqd_comm_init(const void *id_host, int rank, int world, sqd_comm **comm_host) {
    SQD_CHECK_ARG(id_host && comm_host, "sqd_comm_init: null argument");
    SQD_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "sqd_comm_init: rank %d of %d", rank, world);
    if (int rc = bind_rccl(nullptr)) return rc;
    ncclUniqueId id;
    memcpy(id.internal, id_host, SQD_COMM_ID_BYTES);
    ncclComm_t nccl = nullptr;
    SQD_CHECK_RCCL(g_rccl.CommInitRank(&nccl, world, id, rank), "sqd_comm_init", nullptr);
    *comm_host = new sqd_comm{nccl, rank, world};
    return SQD_OK;
}

extern "C" int sqd_comm_rank(const sqd_comm *comm) { return comm ? comm->rank : SQD_EINVAL; }
extern "C" int sqd_comm_world(const sqd_comm *comm) { return comm ? comm->world : SQD_EINVAL; }

// NCCL-style version code of the bound librccl (e.g. 22703 = 2.27.3); 0 if the library does not export ncclGetVersion
extern "C" int sqd_comm_rccl_version(int *version_host) {
    SQD_CHECK_ARG(version_host, "sqd_comm_rccl_version: null argument");
    if (int rc = bind_rccl(nullptr)) return rc;
    *version_host = 0;
    if (g_rccl.GetVersion) SQD_CHECK_RCCL(g_rccl.GetVersion(version_host), "sqd_comm_rccl_version", nullptr);
    return SQD_OK;
}

// ranks RCCL itself counts in the communicator (ncclCommCount): what a bench line records as "ranks joined"
extern "C" int sqd_comm_joined(const sqd_comm *comm, int *count_host) {
    SQD_CHECK_ARG(comm && count_host, "sqd_comm_joined: null argument");
    *count_host = comm->world;
    if (g_rccl.CommCount) SQD_CHECK_RCCL(g_rccl.CommCount(comm->nccl, count_host), "sqd_comm_joined", comm->nccl);
    return SQD_OK;
}

extern "C" int sqd_comm_allreduce(sqd_comm *comm, void *buf, int64_t count, int dtype, int op, void *stream) {
    SQD_CHECK_ARG(comm && (buf || count == 0) && count >= 0, "sqd_comm_allreduce: null argument");
    size_t size;
    int dt = rccl_dtype(dtype, &size);
    SQD_CHECK_ARG(dt >= 0, "sqd_comm_allreduce: dtype %d (0 f32, 1 f64, 2 i32, 3 u8)", dtype);
    SQD_CHECK_ARG(op >= 0 && op <= 3, "sqd_comm_allreduce: op %d (0 sum, 1 avg, 2 max, 3 min)", op);
    SQD_CHECK_ARG(!(op == 1 && dtype >= 2), "sqd_comm_allreduce: average of an integer buffer");
    if (count == 0) return SQD_OK;
    static const int ops[4] = {ncclSum, ncclAvg, ncclMax, ncclMin};
    SQD_CHECK_RCCL(g_rccl.AllReduce(buf, buf, (size_t)count, dt, ops[op], comm->nccl, (hipStream_t)stream), "sqd_comm_allreduce",
                   comm->nccl);
    return SQD_OK;
}

extern "C" int sqd_comm_broadcast(sqd_comm *comm, void *buf, int64_t count, int dtype, int root, void *stream) {
    SQD_CHECK_ARG(comm && (buf || count == 0) && count >= 0, "sqd_comm_broadcast: null argument");
    SQD_CHECK_ARG(root >= 0 && root < comm->world, "sqd_comm_broadcast: root %d of %d", root, comm->world);
    size_t size;
    int dt = rccl_dtype(dtype, &size);
    SQD_CHECK_ARG(dt >= 0, "sqd_comm_broadcast: dtype %d", dtype);
    if (count == 0) return SQD_OK;
    SQD_CHECK_RCCL(g_rccl.Broadcast(buf, buf, (size_t)count, dt, root, comm->nccl, (hipStream_t)stream), "sqd_comm_broadcast",
                   comm->nccl);
    return SQD_OK;
}

extern "C" int sqd_comm_destroy(sqd_comm *comm) {
    if (!comm) return SQD_OK;
    int rc = g_rccl.CommDestroy ? g_rccl.CommDestroy(comm->nccl) : ncclSuccess;
    delete comm;
    if (rc != ncclSuccess) {
        sqd::set_error("sqd_comm_destroy: RCCL error %d", rc);
        return SQD_ECOMM;
    }
    return SQD_OK;
}
