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
 * communication stream (highest priority).  Any time before the first collective; follow it with dvgr_comm_place(). */
int dvgr_comm_create(const void* id, int rank, int world, dvgr_comm** out);
/* Put the communication stream on a hardware queue that shares no command-processor pipe with the streams that are busy
 * during a step.  Why: the command processor serves the process's hardware queues on 4 pipes (a queue's pipe = its creation
 * index mod 4); two non-idle queues of one pipe -- a kernel in flight, or a barrier packet waiting for another queue's signal,
 * which is what the communication stream does for most of a step -- cost EVERY dispatch of the other ~34 us
 * (tools/micro/pipe_probe.hip; a whole-model step went from 4.9 to 9.6 ms that way, rounds 2-3 knew it as "stream creation
 * order matters").  How: up to four candidate streams (consecutive queues = all four pipes) are each held non-idle by a
 * single-wave kernel for 1 ms while a chain of 24 short dispatches is timed on every busy stream, paired with the same chain
 * beside the idle candidate (median of five ratios); the candidate that slows none of them (busy_streams[0] first, then [1]
 * ..., then the library's side and query stream) becomes the communication stream, the others stay idle (destroying a queue
 * re-maps the rest).  ~0.1 s, synchronises the device; call it after every stream the step uses exists and
 * before the first collective.  busy_streams: hipStream_t's of the CALLER (its compute stream first; NULL = the null stream),
 * n_busy <= 5.  report (optional, floats): [candidates, busy streams, chosen, collision mask of the chosen one,
 * undisturbed chain us per busy stream ..., chain us per (candidate, busy stream) ...].  A measurement that fails (a HIP
 * error inside the probe) is NOT an error of this call: the first candidate is installed, the collision mask reads -1 and
 * the call returns 0 -- a rank must never leave the set-up alone because an advisory measurement failed. */
int dvgr_comm_place(dvgr_comm* c, void* const* busy_streams, int n_busy, float* report, int report_len);
int dvgr_comm_destroy(dvgr_comm* c);
int dvgr_comm_world(const dvgr_comm* c);
/* Enqueue, on the communication stream: wait for `wait_event` (hipEvent_t; NULL: for everything `after_stream` holds
 * now), then buf[0 .. count) <- mean over ranks (fp32, in place).  Returns as soon as the work is enqueued.  The library's
 * backward passes both: the event that marks the block final and the stream it was recorded on. */
int dvgr_comm_allreduce_avg(dvgr_comm* c, float* buf, int64_t count, void* wait_event, void* after_stream);
/* In-order mode (on != 0): dvgr_comm_allreduce_avg enqueues on `after_stream` itself (which must then be the stream that
 * produced buf; a non-NULL `wait_event` is waited for on it as well, which costs nothing when it was recorded on that same
 * stream) and dvgr_comm_join does nothing -- no communication stream, no cross-stream event.
 * The collective then delays the later work of its producer stream (the unit's side stream for the layer blocks) but still
 * overlaps the other streams.  On one rank, where a collective is a copy kernel, this is the cheaper form (whole model
 * 1.01 x the plain step against 1.16 x); with several ranks bench.py times both and takes the faster. */
int dvgr_comm_set_inorder(dvgr_comm* c, int on);
/* The same collective enqueued IN ORDER on `stream` (hipStream_t: the stream that produced buf) instead of the communication
 * stream: no event, no second queue; the collective then does not overlap later work of that stream. */
int dvgr_comm_allreduce_avg_on(dvgr_comm* c, float* buf, int64_t count, void* stream);
/* The current device's communication stream (hipStream_t), created on first request. */
int dvgr_comm_device_stream(void** stream_out);
/* `stream` continues only after everything enqueued on the communication stream so far. */
int dvgr_comm_join(dvgr_comm* c, void* stream);

#ifdef __cplusplus
}
#endif
#endif
