// libyue_hip.so -- RCCL over xGMI: communicator set-up and the two collectives of the path (include/yue_hip.h).
#include "host_common.hpp"

#include <rccl/rccl.h>

using yue_host::fail;

#define NCCLCHK(expr)                                                                             \
    do {                                                                                          \
        ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return fail(YUE_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_));        \
    } while (0)

void yue_comm_release(yue_ctx *c) { if (c->comm) (void)ncclCommDestroy(c->comm); c->comm = nullptr; }

namespace yue_host {
// The one data-path collective: the user-factor differences of a group of user blocks, summed over the item
// shards.  In place, fp32, on the given stream (yue_bpr_epoch passes its second stream, event-ordered behind the rounds).
int reduce_user_block(yue_ctx *c, int64_t first, int64_t count, hipStream_t stream) {
    if (c->comm) {
        NCCLCHK(ncclAllReduce(c->dP.p + first, c->dP.p + first, (size_t)count, ncclFloat, ncclSum, c->comm, stream));
        return YUE_OK;
    }
    if (c->seam_reduce) {          // test seam: the same range, summed on the host by the test's callback (blocking)
        c->seam_buf.resize((size_t)count);
        HIPCHK(hipMemcpyAsync(c->seam_buf.data(), c->dP.p + first, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (c->seam_reduce(c->seam_buf.data(), count, 0, c->seam_user)) return fail(YUE_ERR_COMM, "test seam: the reduce callback failed");
        HIPCHK(hipMemcpyAsync(c->dP.p + first, c->seam_buf.data(), (size_t)count * sizeof(float), hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }
    return YUE_OK;
}
}  // namespace yue_host

extern "C" {


int yue_comm_unique_id(void *id128_out) {
    if (!id128_out) return fail(YUE_ERR_ARG, "yue_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) <= YUE_UNIQUE_ID_BYTES, "ncclUniqueId larger than the ABI slot");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    std::memset(id128_out, 0, YUE_UNIQUE_ID_BYTES);
    std::memcpy(id128_out, &id, sizeof id);
    return YUE_OK;
}

int yue_comm_init(yue_ctx *c, const void *id128, int rank, int nranks) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(YUE_ERR_ARG, "yue_comm_init: bad argument");
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    NCCLCHK(ncclCommInitRank(&c->comm, nranks, id, rank));
    c->rank = rank; c->nranks = nranks;
    int ver = 0, cnt = 0;
    NCCLCHK(ncclGetVersion(&ver));
    NCCLCHK(ncclCommCount(c->comm, &cnt));
    c->comm_version = ver; c->comm_nranks_reported = cnt;
    return YUE_OK;
}

#ifdef YUE_TEST_SEAM
// Test build only (make test-seam -> libyue_hip_seam.so; not in include/yue_hip.h, absent from the product library --
// tests/test_abi.py checks): joins the context to a "communicator" of nranks whose collectives are `fn`.
int yue_seam_init(yue_ctx *c, int rank, int nranks, int (*fn)(void *, int64_t, int, void *), void *user) {
    if (!c || !fn || nranks < 1 || rank < 0 || rank >= nranks) return fail(YUE_ERR_ARG, "yue_seam_init: bad argument");
    c->seam_reduce = fn; c->seam_user = user; c->rank = rank; c->nranks = nranks;
    c->comm_nranks_reported = nranks;
    return YUE_OK;
}
#endif

int yue_get_comm_stats(yue_ctx *c, double *allreduce_bytes, int64_t *collectives, double *wait_ms, int *nranks, int *rccl_version) {
    if (!c) return fail(YUE_ERR_ARG, "null context");
    if (allreduce_bytes) *allreduce_bytes = c->comm_bytes;
    if (collectives) *collectives = c->comm_collectives;
    if (wait_ms) *wait_ms = c->comm_wait_ms;
    if (nranks) *nranks = yue_host::on_communicator(c) ? c->comm_nranks_reported : 1;
    if (rccl_version) *rccl_version = c->comm_version;
    return YUE_OK;
}

int yue_allreduce_f64(yue_ctx *c, double *vals, int count) {
    if (!c || !vals || count < 1 || count > 8) return fail(YUE_ERR_ARG, "yue_allreduce_f64: bad argument (count 1..8)");
    if (!c->comm) {
        if (c->seam_reduce && c->seam_reduce(vals, count, 1, c->seam_user)) return fail(YUE_ERR_COMM, "test seam: the reduce callback failed");
        return YUE_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    double *d = c->scal.p + yue_host::kNllSlotsHost;   // scratch scalars (callers read their results before)
    HIPCHK(hipMemcpyAsync(d, vals, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(ncclAllReduce(d, d, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(vals, d, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return YUE_OK;
}

}  // extern "C"
