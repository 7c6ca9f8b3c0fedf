/*
 * dvslam_rccl.h -- C-ABI of libdvslam_rccl.so: the one exchange step of the data-parallel path (SURVEY.md section 8(e),
 * 8(b) `dvs_allreduce_{init,run,destroy}`): a sum all-reduce of the flat fp32 gradient arena over RCCL (xGMI within a node).
 *
 * The reference has no distributed code (vo/train.py builds one optimiser on one device); under data parallelism over
 * frame triplets the gradients of `self.optimizer`'s parameters (vo/train.py:114-117) are summed across ranks between
 * `total_loss.backward()` and `self.optimizer.step()` (vo/train.py:191-192).  The library is separate from
 * libdvslam_hip.so so that single-GPU users never load RCCL.  One process per GPU; the communicator is bound to the
 * device that is current at init time.  Calls are asynchronous on the caller's stream: give the all-reduce a stream of
 * its own (and therefore a hardware queue of its own) and order it with events.
 *
 * Conventions as in dvslam.h: 0 on success, negative on error, dvs_rccl_last_error() for the message.
 */
#ifndef DVSLAM_RCCL_H
#define DVSLAM_RCCL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVS_RCCL_UNIQUE_ID_BYTES 128

typedef struct dvs_comm dvs_comm;   /* opaque communicator */

const char* dvs_rccl_last_error(void);
/* Rank 0 creates the rendezvous token (ncclUniqueId, 128 bytes) and hands it to the other ranks out of band (the Python
 * side broadcasts it over the torch.distributed store it already has). */
int dvs_allreduce_unique_id(void* id_out);
/* Collective over all ranks: communicator of `world_size` ranks on the current HIP device. */
int dvs_allreduce_init(dvs_comm** comm, const void* unique_id, int world_size, int rank);
/* In-place sum of buf[0 .. count) (fp32, device memory) over all ranks, enqueued on `stream` (hipStream_t). */
int dvs_allreduce_run(dvs_comm* comm, float* buf, size_t count, void* stream);
/* Several contiguous ranges of one buffer as ONE RCCL group (one launch): offsets / counts in floats. */
int dvs_allreduce_run_ranges(dvs_comm* comm, float* base, const size_t* offsets, const size_t* counts, int n, void* stream);
int dvs_allreduce_world(const dvs_comm* comm, int* world_size, int* rank);
int dvs_allreduce_destroy(dvs_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* DVSLAM_RCCL_H */
