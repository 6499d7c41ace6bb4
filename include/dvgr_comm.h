/* dvgr_comm.h -- C ABI of the unit's native gradient all-reduce (row e of SURVEY.md section 8: data parallel over
 * independent QA pairs, ONE exchange step: the gradient all-reduce over RCCL / xGMI).
 *
 * Replaces, for the unit's flat gradient buffer, what the reference gets from nn.DataParallel's gather
 * (/root/reference/train.py:80-83): each gradient block (MFB, layer L-1 .. 0, query: dvgr_param_layer_range) is averaged
 * over the ranks by an ncclAllReduce that the BACKWARD CALL ITSELF enqueues -- on the communicator's own high-priority
 * stream, behind the hipEvent that marks the block final -- the moment the block's last kernel has been enqueued
 * (dvgr_opts.comm).  No host callback, no helper thread, no interpreter in the launch path.
 *
 * RCCL is resolved at run time (dvgr_comm_load: dlopen of the librccl the process already uses, e.g. PyTorch's copy), so
 * libdvgr_hip.so has no link-time dependency on it and a process never holds two RCCLs.
 * Plain pointers and sizes; every function returns 0 on success, a negative DVGR_E_* code, or a positive hipError_t /
 * (1000 + ncclResult_t). */
#ifndef DVGR_COMM_H
#define DVGR_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DVGR_COMM_ID_BYTES 128           /* sizeof(ncclUniqueId) */
typedef struct dvgr_comm dvgr_comm;

/* Resolve the RCCL entry points from `librccl_path` (NULL: "librccl.so.1" by the loader's search path).  Idempotent. */
int dvgr_comm_load(const char* librccl_path);
/* Rank 0: a fresh unique id (DVGR_COMM_ID_BYTES bytes) to hand to every rank (the host side broadcasts it over whatever
 * channel it has: torch.distributed, MPI, a file). */
int dvgr_comm_unique_id(void* id_out);
/* Collective over all ranks: the communicator of `rank` in a job of `world` ranks on the CURRENT device, plus the device's
 * communication stream (highest priority).  Must be called BEFORE the unit's first forward / backward on this device
 * (DVGR_E_ORDER otherwise): the runtime maps streams to hardware queues in creation order, and a communication stream
 * created after the unit's side stream made a data-parallel step 3 x slower (csrc/dvgr_side.h). */
int dvgr_comm_create(const void* id, int rank, int world, dvgr_comm** out);
int dvgr_comm_destroy(dvgr_comm* c);
int dvgr_comm_world(const dvgr_comm* c);
/* Enqueue, on the communication stream: wait for `wait_event` (hipEvent_t; NULL: for everything `after_stream` holds
 * now), then buf[0 .. count) <- mean over ranks (fp32, in place).  Returns as soon as the work is enqueued. */
int dvgr_comm_allreduce_avg(dvgr_comm* c, float* buf, int64_t count, void* wait_event, void* after_stream);
/* The current device's communication stream (hipStream_t), created on first request under the same ordering rule as
 * dvgr_comm_create. */
int dvgr_comm_device_stream(void** stream_out);
/* `stream` continues only after everything enqueued on the communication stream so far. */
int dvgr_comm_join(dvgr_comm* c, void* stream);

#ifdef __cplusplus
}
#endif
#endif
