// coll.hip -- Coll-1: the one collective of the hot path, an all-reduce of [sum a, sum b, sum c, sum d, n_valid]
// (5 doubles) over the workers' refined planes, i.e. np.nanmean(planes.txt) of
// gridding/wassgridsurface/wassgridsurface.py:672-678 computed without a shared file.  RCCL over xGMI, one rank per
// GPU; 40 bytes, pure latency.
//
// librccl is opened on first use (dlopen) instead of being linked: a process that also holds PyTorch's bundled copy
// of RCCL (bench.py, the tests) must not see two sets of nccl* symbols, and single-GPU users need no RCCL at all.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace wass {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return GetUniqueId && CommInitRank && AllReduce && CommDestroy; }
};

static Rccl& rccl()
{
    static Rccl r = [] {
        Rccl x;
        const char* env = getenv("WASS_RCCL_LIB");
        const char* names[] = { env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names) {
            if (!n) continue;
            x.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (x.so) break;
        }
        if (x.so) {
            x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.so, "ncclGetUniqueId");
            x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.so, "ncclCommInitRank");
            x.AllReduce = (decltype(x.AllReduce))dlsym(x.so, "ncclAllReduce");
            x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.so, "ncclCommDestroy");
            x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.so, "ncclGetErrorString");
        }
        return x;
    }();
    return r;
}

static int rccl_err(wass_ctx* c, const char* what, ncclResult_t e)
{
    return set_err(c, WASS_ERR_DEVICE, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error");
}

void coll_release(wass_ctx* c)
{
    if (c->coll_comm && rccl().CommDestroy) (void)rccl().CommDestroy((ncclComm_t)c->coll_comm);
    c->coll_comm = nullptr;
    if (c->coll_buf) (void)hipFree(c->coll_buf);
    c->coll_buf = nullptr;
}

}  // namespace wass

using namespace wass;

extern "C" {

int wass_coll_unique_id(unsigned char id_out[128])
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!id_out || !rccl().ok()) return WASS_ERR_DEVICE;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return WASS_ERR_DEVICE;
    memcpy(id_out, &id, sizeof id);
    return WASS_OK;
}

int wass_coll_init(wass_ctx* c, int rank, int world, const unsigned char id[128])
{
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return set_err(c, WASS_ERR_INVALID_ARG, "bad collective geometry");
    if (!rccl().ok()) return set_err(c, WASS_ERR_DEVICE, "librccl could not be loaded");
    WASS_HIP(c, hipSetDevice(c->device));
    coll_release(c);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t e = rccl().CommInitRank(&comm, world, uid, rank);
    if (e != ncclSuccess) return rccl_err(c, "ncclCommInitRank", e);
    c->coll_comm = comm;
    c->coll_world = world;
    WASS_HIP(c, hipMalloc(&c->coll_buf, 64 * sizeof(double)));
    return WASS_OK;
}

int wass_coll_allreduce_sum_f64(wass_ctx* c, double* values, int count)
{
    if (!c || !values || count < 1 || count > 64) return set_err(c, WASS_ERR_INVALID_ARG, "bad all-reduce arguments");
    if (!c->coll_comm) return set_err(c, WASS_ERR_INVALID_ARG, "wass_coll_init has not been called");
    WASS_HIP(c, hipSetDevice(c->device));
    WASS_HIP(c, hipMemcpyAsync(c->coll_buf, values, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const ncclResult_t e = rccl().AllReduce(c->coll_buf, c->coll_buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)c->coll_comm, c->stream);
    if (e != ncclSuccess) return rccl_err(c, "ncclAllReduce", e);
    WASS_HIP(c, hipMemcpyAsync(values, c->coll_buf, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    return WASS_OK;
}

}  // extern "C"
