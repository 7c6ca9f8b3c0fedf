// libdvslam_rccl.so: RCCL sum all-reduce of the flat gradient arena (include/dvslam_rccl.h).
#include "../../../include/dvslam_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

struct dvs_comm {
    ncclComm_t comm;
    int world, rank;
};

namespace {
char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
#define RCCL_TRY(call, what)                                                                  \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess) return fail(-2, "%s: %s", what, ncclGetErrorString(r_));       \
    } while (0)
static_assert(sizeof(ncclUniqueId) <= DVS_RCCL_UNIQUE_ID_BYTES, "unique id does not fit the ABI's token");
}  // namespace

extern "C" {

const char* dvs_rccl_last_error(void) { return err_buf(); }

int dvs_allreduce_unique_id(void* id_out) {
    if (!id_out) return fail(-1, "dvs_allreduce_unique_id: null output");
    ncclUniqueId id;
    RCCL_TRY(ncclGetUniqueId(&id), "ncclGetUniqueId");
    std::memset(id_out, 0, DVS_RCCL_UNIQUE_ID_BYTES);
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

int dvs_allreduce_init(dvs_comm** comm, const void* unique_id, int world_size, int rank) {
    if (!comm || !unique_id || world_size < 1 || rank < 0 || rank >= world_size)
        return fail(-1, "dvs_allreduce_init: bad argument (world %d, rank %d)", world_size, rank);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c;
    RCCL_TRY(ncclCommInitRank(&c, world_size, id, rank), "ncclCommInitRank");
    *comm = new dvs_comm{c, world_size, rank};
    return 0;
}

int dvs_allreduce_run(dvs_comm* comm, float* buf, size_t count, void* stream) {
    if (!comm || !buf) return fail(-1, "dvs_allreduce_run: null argument");
    if (count == 0) return 0;
    RCCL_TRY(ncclAllReduce(buf, buf, count, ncclFloat32, ncclSum, comm->comm, static_cast<hipStream_t>(stream)), "ncclAllReduce");
    return 0;
}

int dvs_allreduce_run_ranges(dvs_comm* comm, float* base, const size_t* offsets, const size_t* counts, int n, void* stream) {
    if (!comm || !base || !offsets || !counts || n < 0) return fail(-1, "dvs_allreduce_run_ranges: bad argument");
    RCCL_TRY(ncclGroupStart(), "ncclGroupStart");
    for (int i = 0; i < n; ++i) {
        if (counts[i] == 0) continue;
        ncclResult_t r = ncclAllReduce(base + offsets[i], base + offsets[i], counts[i], ncclFloat32, ncclSum, comm->comm,
                                       static_cast<hipStream_t>(stream));
        if (r != ncclSuccess) {
            (void)ncclGroupEnd();
            return fail(-2, "ncclAllReduce (range %d): %s", i, ncclGetErrorString(r));
        }
    }
    RCCL_TRY(ncclGroupEnd(), "ncclGroupEnd");
    return 0;
}

int dvs_allreduce_world(const dvs_comm* comm, int* world_size, int* rank) {
    if (!comm) return fail(-1, "dvs_allreduce_world: null communicator");
    if (world_size) *world_size = comm->world;
    if (rank) *rank = comm->rank;
    return 0;
}

int dvs_allreduce_destroy(dvs_comm* comm) {
    if (!comm) return 0;
    ncclResult_t r = ncclCommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) return fail(-2, "ncclCommDestroy: %s", ncclGetErrorString(r));
    return 0;
}

}  // extern "C"
