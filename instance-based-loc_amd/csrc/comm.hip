// comm.hip -- the two collectives of the sharded path on RCCL (xGMI inside a node), behind the C-ABI, for hosts that do not go
// through torch.distributed: all-gather of the per-shard candidate lists (north star: "RCCL all-gather of per-shard top-k matches
// over xGMI before registration") and all-reduce(MIN) of per-point nearest distances for the whole-memory evaluation with sharded
// clouds.  The reference issues no collective on this path (it is single-process); these are new.
// One communicator per process / GPU; the unique id is created on rank 0 and handed to the other ranks by the host (any channel:
// the Python layer broadcasts it with torch.distributed, ibloc_amd.parallel.RcclComm).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>

#include "ibl_common.h"
#include "ibloc.h"

struct ibl_comm {
    ncclComm_t comm;
    int rank, world;
};

#define IBL_NCCL_CHECK(expr)                                                                                      \
    do {                                                                                                          \
        ncclResult_t _r = (expr);                                                                                 \
        if (_r != ncclSuccess) return ibl_set_error(IBL_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(_r)); \
    } while (0)

extern "C" int ibl_comm_unique_id(void* out, int out_bytes) {
    if (!out || out_bytes < (int)sizeof(ncclUniqueId)) return ibl_set_error(IBL_ERR_ARG, "ibl_comm_unique_id: need %d bytes", (int)sizeof(ncclUniqueId));
    ncclUniqueId id;
    IBL_NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return (int)sizeof(ncclUniqueId);
}

extern "C" int ibl_comm_init(ibl_comm** out, int rank, int world, const void* unique_id, int id_bytes) {
    if (!out || !unique_id || world <= 0 || rank < 0 || rank >= world || id_bytes < (int)sizeof(ncclUniqueId))
        return ibl_set_error(IBL_ERR_ARG, "ibl_comm_init: bad argument");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ibl_comm* c = new ibl_comm();
    c->rank = rank; c->world = world;
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);        // uses the calling thread's current device
    if (r != ncclSuccess) { delete c; return ibl_set_error(IBL_ERR_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r)); }
    *out = c;
    return IBL_OK;
}

extern "C" int ibl_comm_destroy(ibl_comm* c) {
    if (!c) return IBL_OK;
    (void)ncclCommDestroy(c->comm);
    delete c;
    return IBL_OK;
}

extern "C" int ibl_allgather_topk(ibl_comm* c, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
    if (!c || !send || !recv || bytes_per_rank < 0) return ibl_set_error(IBL_ERR_ARG, "ibl_allgather_topk: bad argument");
    if (bytes_per_rank == 0) return IBL_OK;
    IBL_NCCL_CHECK(ncclAllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, c->comm, (hipStream_t)stream));
    return IBL_OK;
}

// equal-block all-to-all (ncclSend / ncclRecv pairs in one group): block r of `send` goes to rank r, block r of `recv` comes from it.
// The per-shard candidate lists of the sharded match: each rank ends up with the W lists of ITS OWN query rows only.
extern "C" int ibl_alltoall(ibl_comm* c, const void* send, void* recv, int64_t bytes_per_pair, void* stream) {
    if (!c || !send || !recv || bytes_per_pair < 0) return ibl_set_error(IBL_ERR_ARG, "ibl_alltoall: bad argument");
    if (bytes_per_pair == 0) return IBL_OK;
    IBL_NCCL_CHECK(ncclGroupStart());
    for (int r = 0; r < c->world; ++r) {
        IBL_NCCL_CHECK(ncclSend(reinterpret_cast<const char*>(send) + (size_t)r * bytes_per_pair, (size_t)bytes_per_pair, ncclInt8, r, c->comm,
                                (hipStream_t)stream));
        IBL_NCCL_CHECK(ncclRecv(reinterpret_cast<char*>(recv) + (size_t)r * bytes_per_pair, (size_t)bytes_per_pair, ncclInt8, r, c->comm,
                                (hipStream_t)stream));
    }
    IBL_NCCL_CHECK(ncclGroupEnd());
    return IBL_OK;
}

extern "C" int ibl_allreduce_min(ibl_comm* c, float* buf, int64_t n, void* stream) {
    if (!c || !buf || n < 0) return ibl_set_error(IBL_ERR_ARG, "ibl_allreduce_min: bad argument");
    if (n == 0) return IBL_OK;
    IBL_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)n, ncclFloat, ncclMin, c->comm, (hipStream_t)stream));
    return IBL_OK;
}

extern "C" int ibl_allreduce_max_i32(ibl_comm* c, int32_t* buf, int64_t n, void* stream) {
    if (!c || !buf || n < 0) return ibl_set_error(IBL_ERR_ARG, "ibl_allreduce_max_i32: bad argument");
    if (n == 0) return IBL_OK;
    IBL_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)n, ncclInt32, ncclMax, c->comm, (hipStream_t)stream));
    return IBL_OK;
}
