/*
 * dvgr.h -- C ABI of the MI355X-native DualVGR reasoning unit (libdvgr_hip.so).
 *
 * The reference (NJUPT-MCC/DualVGR-VideoQA) has no FFI or plugin layer: the hot path sits behind a
 * PyTorch nn.Module contract.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference repository).  The Python host side
 * (dualvgr-videoqa_amd/unit.py) binds these with ctypes and exposes the reference's own
 * `DualVGRUnit_multiple` constructor / forward() signature (model/models.py:87,121).
 *
 * Conventions
 *   - every tensor is fp32, contiguous, row-major, last dimension fastest, resident in HBM;
 *   - the library never allocates, frees or retains device memory and never synchronises the
 *     device: it only enqueues kernels on the stream it is given and, for work off the critical path, on its internal
 *     side stream, which forks from and joins back into the given stream inside the same call (the caller sees ordinary
 *     single-stream semantics; the fork / join is capturable by hipGraph stream capture);
 *   - functions return 0 on success, a negative DVGR_E_* code for invalid arguments, or the
 *     positive hipError_t value of a failed launch;
 *   - calls are re-entrant and thread-safe (autograd calls backward from its own thread).  The library keeps exactly
 *     one piece of process-wide state: per device, its internal lower-priority side / query streams (created on first use
 *     or by dvgr_streams_place -- the one place where a call synchronises the device: their hardware queues are picked by
 *     a ~0.1 s measurement, see dvgr_streams_place below; shared by all threads, never destroyed) plus a per-thread
 *     pool of fork / join events on them.
 *     It holds no pointer to caller memory between calls; DVGR_OPT_SINGLE_STREAM makes a call use neither.
 */
#ifndef DVGR_H
#define DVGR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVGR_F32 0  /* fp32 operands on v_mfma_f32_16x16x4_f32 (exact fp32; parity mode)           */
#define DVGR_BF16 1 /* bf16 operands (rounded on chip) on v_mfma_f32_16x16x32_bf16, fp32 accumulate */
#define DVGR_BF16X3 2 /* every fp32 operand split as hi + lo bf16; 3 bf16 MFMAs per product (hi.hi + hi.lo +
                         lo.hi), fp32 accumulate: fp32-grade results (~1e-5 relative) at 3/16 the MFMA cost */

#define DVGR_E_DIMS (-1)      /* unsupported dimension (see dvgr_check_dims)            */
#define DVGR_E_NULL (-2)      /* a required pointer is NULL                             */
#define DVGR_E_WORKSPACE (-3) /* saved/workspace buffer too small                       */
#define DVGR_E_ALIGN (-4)     /* a pointer is not 16-byte aligned                       */
#define DVGR_E_KIND (-5)      /* unknown parameter / saved-tensor kind                  */
#define DVGR_E_ORDER (-6)     /* (rounds 2-3: communicator created after the unit ran; no longer returned -- dvgr_comm_place) */

/* Problem dimensions.  Mirrors the constructor arguments of DualVGRUnit_multiple
 * (model/models.py:87) plus the run-time batch shape of its forward (model/models.py:121-133). */
typedef struct dvgr_dims {
  int32_t B;     /* batch size (QA pairs)                                            */
  int32_t N;     /* num_of_nodes = clips per video (configs yml: train.num_of_nodes) */
  int32_t T;     /* padded question length of this batch                             */
  int32_t D;     /* module_dim (768); multiple of 64                                 */
  int32_t Wd;    /* word_dim (300); multiple of 4                                    */
  int32_t L;     /* unit_layers                                                      */
  int32_t G;     /* graph_layers, 1 .. 4 (all reference configs use 1).  G > 1 follows model/models.py:150-160: unit layer i,
                    graph layer j uses GAT module [i + j] of each family (shared between unit layers; modules > L + G - 2 exist
                    in the state dict only), every list output has L * G entries (entry i * G + j), the backward runs on the
                    caller's stream alone and refuses layer_done_events / comm (no block is final before the end) */
  int32_t dtype; /* DVGR_F32, DVGR_BF16 or DVGR_BF16X3 (precision of the MFMA operands) */
} dvgr_dims;

/* Kinds of parameter tensors inside the flat parameter buffer.  `mod` = 0 appearance, 1 motion.
 * Reference state_dict names each kind aliases:
 *   QA_W1/B1  queryAttn.l.feat_enhance.{weight,bias}      QA_W2/B2  queryAttn.l.fc.{weight,bias}
 *   WQ/BQ     queryPunish_{appear,motion}.l.query_weight.{weight,bias}
 *   GAT_W     rows [h*F,(h+1)*F) = {acGCN|mcGCN}.l.attention_h.W.weight for h<4 (common GAT),
 *             rows [(4+h)*F, ...) = {appearance_GCN|motion_GCN}.l.attention_h.W.weight (specific)
 *   GAT_B     same packing of the W.bias vectors;  GAT_A  row h = a.weight (1, 2F);  GAT_C  a.bias
 *   SF_P1/BP  attention_{appearance,motion}.l.project.0.{weight,bias}   SF_P2  project.2.weight
 *   MFB_*     visualfusion.{linear0,linear1,linear_out}.{weight,bias}  (layer/mod ignored)      */
enum dvgr_param_kind {
  DVGR_P_QA_W1 = 0, DVGR_P_QA_B1, DVGR_P_QA_W2, DVGR_P_QA_B2,
  DVGR_P_WQ, DVGR_P_BQ,
  DVGR_P_GAT_W, DVGR_P_GAT_B, DVGR_P_GAT_A, DVGR_P_GAT_C,
  DVGR_P_SF_P1, DVGR_P_SF_BP, DVGR_P_SF_P2,
  DVGR_P_MFB_W0, DVGR_P_MFB_B0, DVGR_P_MFB_W1, DVGR_P_MFB_B1, DVGR_P_MFB_WO, DVGR_P_MFB_BO,
  DVGR_P_KINDS
};

/* Kinds of tensors the forward saves for the backward (introspection for the parity tests). */
enum dvgr_saved_kind {
  DVGR_S_QA_D = 0,   /* (B*T, D)  L2-normalised feat_enhance output                 per layer     */
  DVGR_S_QA_NRM,     /* (B*T)     max(||y||, 1e-12)                                                */
  DVGR_S_QA_A,       /* (B*T)     softmax over all T                                               */
  DVGR_S_QA_AH,      /* (B*T)     masked + renormalised attention                                  */
  DVGR_S_QA_S,       /* (B)       sum of masked attention + 1e-5                                   */
  DVGR_S_QA_Q,       /* (B, Wd)   cycle question vector                                            */
  DVGR_S_P,          /* (B, D)    query_weight(q)                                   per layer, mod */
  DVGR_S_S,          /* (B*N)     punish gate sigmoid(X.p)                                         */
  DVGR_S_X,          /* (B*N, D)  layer OUTPUT X' = X + o (input of layer l+1 / of the MFB)        */
  DVGR_S_WH,         /* (B*N, 8F) packed head projections                                          */
  DVGR_S_U,          /* (B, 8, N) a_l . Wh_i                                                       */
  DVGR_S_V,          /* (B, 8, N) a_r . Wh_j                                                       */
  DVGR_S_TN,         /* (2, B*N, D) tanh(project.0(z_v)), v = 0 common, 1 specific                 */
  DVGR_S_BETA,       /* (B*N, 2)  softmax over the two views                                       */
  DVGR_S_MFB_X0,     /* (B*N, 512) ELU(linear0(A))                                  once           */
  DVGR_S_MFB_X1,     /* (B*N, 512) ELU(linear1(M))                                                 */
  DVGR_S_MFB_Z,      /* (B*N, 256) pair-summed product                                             */
  DVGR_S_XD,         /* (2, B*N, D) layer input after the x-site dropout of the common / specific GAT
                        (training mode only).  In the bf16 modes with B*N % 32 == 0 and B*N > 256 only the
                        bf16 planes of this slot are written (its readers are contractions), the fp32
                        values are not                                                 per layer, mod */
  DVGR_S_KINDS
};

/* Inputs and outputs of DualVGRUnit_multiple.forward (model/models.py:121-173). */
typedef struct dvgr_io {
  const float *A;       /* appearance_video_feat       (B, N, D)  */
  const float *M;       /* motion_video_feat           (B, N, D)  */
  const float *dq;      /* dynamic_question_embedding  (B, T, D)  */
  const float *words;   /* word_embedding              (B, T, Wd) */
  const int64_t *qlen;  /* question_len                (B) int64  */
  float *vis;           /* visual_feature              (B, N, D)  */
  float *aq_embed;      /* last layer's fused appearance embedding (B, N, D) */
  float *mq_embed;      /* last layer's fused motion embedding     (B, N, D) */
  float *com_app;       /* com_app_list    as one (L * G, B, N, D) tensor */
  float *com_mot;       /* com_motion_list as one (L * G, B, N, D) tensor */
  float *aq_fusion;     /* aq_fusion_list  as one (L * G, B, N, D) tensor */
  float *mq_fusion;     /* mq_fusion_list  as one (L * G, B, N, D) tensor */
} dvgr_io;

/* Upstream gradients (any may be NULL = zero) and gradient outputs of the backward pass. */
typedef struct dvgr_grads {
  const float *g_vis, *g_aq_embed, *g_mq_embed;                 /* (B, N, D)    */
  const float *g_com_app, *g_com_mot, *g_aq_fusion, *g_mq_fusion; /* (L * G, B, N, D) */
  float *gA, *gM;       /* (B, N, D)  */
  float *gdq;           /* (B, T, D)  */
  float *gwords;        /* (B, T, Wd) */
  float *gparams;       /* flat, same layout as the parameter buffer; every element is written */
} dvgr_grads;

/* Library version string "dvgr-hip <semver> gfx950". */
const char *dvgr_version(void);
/* Creates, for the calling thread's current device, the two internal lower-priority streams of the two-stream schedules
 * (side, query; otherwise created by the first forward / backward) -- each on the hardware queue that disturbs
 * `caller_stream` (hipStream_t, the stream the unit will be called on; NULL = the null stream) and the streams created before
 * it least.  Background (csrc/dvgr_place.hip, tools/micro/pipe_probe.hip): the command processor serves a process's hardware
 * queues on 4 pipes, and two busy queues of ONE pipe cost every dispatch of both ~25 us; which queues share a pipe follows
 * creation order, so the library measures instead of assuming an order (~20-100 ms per stream, once; synchronises the device).
 * HOSTS SHOULD CALL THIS DURING SET-UP (the Python host side does, when the module moves to its device): left to the first
 * forward / backward call, the measurement runs inside that call -- a device synchronise plus ~0.1 s of probe kernels on the
 * caller's stream.  While a stream capture is active (on the caller's stream, or a global-mode capture anywhere) the probe is
 * skipped and the streams are created plainly; at most 12 candidate streams are ever created per device (the losers of one
 * placement are the first candidates of the next).
 * DVGR_NO_STREAM_PLACEMENT=1 in the environment: plain stream creation. */
int dvgr_streams_place(void *caller_stream);
int dvgr_streams_init(void);                    /* = dvgr_streams_place(NULL) */
/* A stream for the HOST side's own concurrent work (an encoder stream, a loader's copy stream), placed the same way against
 * `caller_stream` and every stream placed before.  priority_class: -1 high, 0 normal, 1 low; rank: 2 .. 9, lower = more
 * important when a later stream has to share a pipe with one of them; report (optional): as dvgr_comm_place.  The stream is
 * never destroyed by the library; wrap it (torch.cuda.ExternalStream). */
int dvgr_stream_create_placed(int priority_class, int rank, void *caller_stream, void **stream_out, float *report, int report_len);

/* 0 when `d` is supported: B,N,T,L >= 1, 1 <= G <= 4, D % 64 == 0, Wd % 4 == 0, N <= 64, T <= 256. */
int dvgr_check_dims(const dvgr_dims *d);

/* Number of fp32 elements of the flat parameter buffer (replaces the nn.ModuleList storage built
 * by DualVGRUnit_multiple.__init__, model/models.py:87-112). */
int64_t dvgr_param_count(const dvgr_dims *d);

/* Element offset and 2-D shape (rows, cols; vectors have rows = 1) of one parameter tensor inside
 * the flat buffer.  Every offset is a multiple of 64 elements. */
int dvgr_param_offset(const dvgr_dims *d, int kind, int layer, int mod,
                      int64_t *offset, int64_t *rows, int64_t *cols);

/* Bytes of the `saved` buffer: an fp32 region the forward writes for the backward (dvgr_saved_offset gives
 * the byte offset of one tensor inside it) followed by a region of bf16 hi / lo planes of the same size. */
size_t dvgr_saved_bytes(const dvgr_dims *d);
int dvgr_saved_offset(const dvgr_dims *d, int kind, int layer, int mod, size_t *offset_bytes);

/* Scratch bytes needed by forward / backward (contents undefined between calls); fp32 region + planes region. */
size_t dvgr_workspace_bytes(const dvgr_dims *d);

/* bf16 hi / lo planes of the parameters for DVGR_BF16 / DVGR_BF16X3: hi = bf16(w) at element offset o,
 * lo = bf16(w - hi) at dvgr_param_count() + o (same offsets as the fp32 buffer).  Call after every
 * optimizer step.  `params_lp` must hold 2 * dvgr_param_count() * 2 bytes. */
int dvgr_pack_weights(const dvgr_dims *d, const float *params, void *params_lp, void *stream);
/* Transposed planes of the SF attention's project[0] weights (model/Attention.py:15) for the fused GAT-view backward
 * (dvgr_opts.p1t): `p1t` receives dvgr_p1t_elems(d) bf16 -- hi planes of (layer, modality) = (0,0), (0,1), (1,0) ... each
 * (D, D) with out[i][o] = P1[o][i], then the lo planes in the same order -- read from the planes `params_lp` holds (call
 * it after dvgr_pack_weights / after an optimizer step that rewrote the planes).  DVGR_E_KIND for DVGR_F32 dims. */
size_t dvgr_p1t_elems(const dvgr_dims *d);
/* 1 when dvgr_unit_backward at these dimensions and dvgr_opts.flags takes the fused GAT-view backward, i.e. reads P1^T
 * (dvgr_opts.p1t, or its own per-call transposition); 0 otherwise -- a host prepares the transposed planes only then. */
int dvgr_backward_uses_p1t(const dvgr_dims *d, int32_t flags);
/* 8-bit block-scaled lo planes of the weight operands of the fused GAT-view kernels (dvgr_opts.lo8; DVGR_BF16X3 only): per
 * row, s = max |lo| / 127 and q = round(lo / s) + 128, where lo is the bf16 lo plane (the 9th .. 16th significant bit of the
 * weight) -- the kernels rebuild lo ~ (q - 128) s in registers.  `lo8` receives dvgr_lo8_bytes(d) bytes: Wcat of every (layer,
 * modality) from `params_lp`, P1^T of every (layer, modality) from `p1t` (dvgr_pack_p1t; call that first), then the scales.
 * Call it whenever the planes were rewritten.  DVGR_E_KIND unless d->dtype == DVGR_BF16X3. */
size_t dvgr_lo8_bytes(const dvgr_dims *d);
int dvgr_pack_lo8(const dvgr_dims *d, const void *params_lp, const void *p1t, void *lo8, void *stream);
int dvgr_pack_p1t(const dvgr_dims *d, const void *params_lp, void *p1t, void *stream);

/* Launch sites a caller can bracket with a pair of hipEvents (dvgr_opts.probe): the events are
 * recorded on the launch stream immediately before / after that one kernel launch, which is how
 * bench.py measures the dominant kernel's duration live inside the timed region. */
enum dvgr_probe_site {
  DVGR_PROBE_NONE = 0,
  DVGR_PROBE_FWD_QA_GEMM,    /* S0  feat_enhance of all layers            (forward)  */
  DVGR_PROBE_FWD_GAT_GEMM,   /* S2  Wh = X . Wcat^T, both modalities      (forward)  */
  DVGR_PROBE_FWD_GAT_ATTN,   /* S3  attention + aggregation               (forward)  */
  DVGR_PROBE_FWD_SF_GEMM,    /* S4  tanh(z . P1^T), 2 views x 2 modalities (forward) */
  DVGR_PROBE_FWD_MFB_GEMM,   /* S6  linear0 / linear1                     (forward)  */
  DVGR_PROBE_BWD_SF_DW,      /* (merged into DVGR_PROBE_BWD_GAT_DW: one launch per layer)     */
  DVGR_PROBE_BWD_SF_DX,      /* dz += dpre . P1                           (backward) */
  DVGR_PROBE_BWD_GAT_ATTN,   /* attention backward                        (backward) */
  DVGR_PROBE_BWD_GAT_DW,     /* dWcat = dWh^T . X and dP1 = dpre^T . z    (backward) */
  DVGR_PROBE_BWD_GAT_DX,     /* dX += dWh . Wcat                          (backward) */
  DVGR_PROBE_SITES
};

/* dvgr_opts.flags.  By default forward and backward enqueue what is off their critical path (forward: the query-side
 * chain S0 and, with dvgr_opts.loss, the loss kernels; backward: dP1, the query branch, every column sum) on an internal
 * lower-priority side stream (one per device and calling thread, created on first use) so that it overlaps the
 * latency-bound kernels of the critical path; the side stream forks from and joins back into `stream` inside the call,
 * so the caller sees ordinary single-stream semantics.  DVGR_OPT_SINGLE_STREAM keeps every launch on `stream`. */
#define DVGR_OPT_SINGLE_STREAM 1
/* The backward additionally runs its query branch (S1 tail and S0 backward: a chain of small, latency-bound kernels) on a
 * second internal stream, created on first use.  DVGR_OPT_NO_QUERY_STREAM keeps that branch on the side stream and never
 * creates the second one: set it when the process drives further streams of its own beside the call (a gradient
 * reducer's: with this stream present as well a data-parallel step took 3.5 ms instead of 1.4 ms on ROCm 7.0 -- the
 * runtime then spends the time on the host between launches; dualvgr-videoqa_amd/unit.py sets it whenever layer events
 * are installed).  The environment variable DVGR_NO_QUERY_STREAM=1 has the same effect for every call. */
#define DVGR_OPT_NO_QUERY_STREAM 2
/* Forward: the GAT projection (S2) and the attention (S3) of a layer run as ONE launch (csrc/dvgr_gatf.hip: the attention is
 * the contraction's epilogue on a tile of whole samples x one head) wherever the shape allows it -- bf16 modes, D = 768 --
 * and the punish gate does not depend on work the separate projection would hide (layer 0 beside the side stream keeps two
 * launches).  DVGR_OPT_NO_FUSED_GAT forces the two-launch form everywhere (A/B timing, parity of the two forms). */
#define DVGR_OPT_NO_FUSED_GAT 4
/* (The same bit also keeps S6 -- linear0 / linear1 + ELU + the pooled product, one launch on the planes path, k_mfb_fwd --
 * in its two-launch form: the flag means "no fused forward kernels".) */
/* Backward with a query stream of its own: the d Wq / d bq leaves of the query block (needing only d p) are enqueued on the
 * side stream beside layer 0's query chain instead of behind it.  DVGR_OPT_LEAVES_ON_QUERY keeps them on the query stream
 * (A/B timing). */
#define DVGR_OPT_LEAVES_ON_QUERY 8
/* Backward: the S4 data contraction dz_v += dpre_v . P1 and the attention backward (S3) of a layer run as ONE launch
 * (csrc/dvgr_gatb.hip: the attention backward is the contraction's epilogue on a tile of whole samples x one head's columns;
 * model/Attention.py:20-23 and model/GraphNN.py:95-111 backward) wherever the shape allows it -- bf16 modes, D = 768,
 * graph_layers = 1, clip counts whose per-sample images fit the compute unit's LDS (N <= 20 at 4-5 row tiles).
 * DVGR_OPT_NO_FUSED_GAT_BWD keeps the two-launch form (A/B timing, parity of the two forms); DVGR_OPT_NO_FUSED_GAT implies it. */
#define DVGR_OPT_NO_FUSED_GAT_BWD 16
/* The fused backward form is taken where it is supported AND measured faster (csrc/dvgr_gatb.hip gat_fused_bwd_preferred: launches
 * of >= 200 workgroups, or host-bound small batches); DVGR_OPT_FORCE_FUSED_GAT_BWD takes it wherever it is supported (tests, A/B). */
#define DVGR_OPT_FORCE_FUSED_GAT_BWD 32
/* The fused backward launch also forms the part of dz that the S5 / S4 row kernel (k_sf_bwd) would otherwise store and it would
 * read back -- beta_v (dX' + g_embed) + the list gradient -- from the tensors themselves.  DVGR_OPT_FUSED_BWD_READS_DZ keeps the
 * row kernel's store and the read (A/B timing, parity of the two forms). */
#define DVGR_OPT_FUSED_BWD_READS_DZ 64
/* Forward: the S4 contraction's epilogue hands S5 the logits tanh(z_v P1^T + b) . p2 as partial dots per 32-column block
 * (GemmProblem::rowdot_out) where the launch runs on whole tiles of the planes kernels, so that S5 does not read the tanh
 * outputs back.  DVGR_OPT_S5_READS_TN keeps S5's own dot products over them (A/B timing, parity of the two forms). */
#define DVGR_OPT_S5_READS_TN 128
/* The fused GAT-view kernels stream the bf16 lo planes of their weight operands even when dvgr_opts.lo8 is set (A/B timing,
 * parity of the two forms). */
#define DVGR_OPT_NO_LO8 256
/* The fused GAT-view kernels run their K loops on 64-deep K-tiles where that form is built (bf16x3, 4- and 5-row-tile shapes):
 * a row piece is then a whole 128-byte line, half the L2 requests per byte, on a 2-slot ring (csrc/dvgr_gatf.h, KB).
 * DVGR_OPT_KB32 keeps the 32-deep tiles of round 4 (A/B timing; also what dvgr_opts.lo8 applies to). */
#define DVGR_OPT_KB32 512
/* Where a launch of the fused GAT-view kernels on 64-row tiles (N <= 16) would leave most of the chip idle they run on 32-row
 * tiles -- twice the workgroups (csrc/dvgr_gatf.h, gatf_half_shape; round 6: MSRVTT-QA's per-GPU batch of 32 x 16 clips).
 * DVGR_OPT_NO_HALF_GAT_TILES keeps the 64-row tiles (A/B timing, parity of the two forms). */
#define DVGR_OPT_NO_HALF_GAT_TILES 1024
/* The backward completes the running gradient of a layer boundary -- dX += dtau p + the specific GAT's partial, the dX half of
 * the S1 backward (model/utils.py:92-105) -- inside the next layer's row pass (k_sf_bwd) instead of by a kernel of its own
 * (layers >= 1, graph_layers 1; round 6).  DVGR_OPT_NO_FOLDED_PUNISH keeps the separate launch (A/B timing, parity of the two forms). */
#define DVGR_OPT_NO_FOLDED_PUNISH 2048
/* Where the forward runs the query side S0 (contraction Y = dq W1^T, row kernels, Wq contraction) when it has its side stream.
 * Default (round 6): LAYER 0's share on the caller's stream in front of layer 0 -- which then takes the fused GAT-view kernel: its
 * gate exists before it starts, no fork / join on its path -- and the other layers' share on the side stream, needed by layer 0's
 * S5.  DVGR_OPT_S0_ON_MAIN: all layers' share on the caller's stream.  DVGR_OPT_S0_ON_SIDE: the form of rounds 1-5 (contraction on
 * the caller's stream, rows + Wq on the side stream under layer 0's separate projection, joined in front of the gate). */
#define DVGR_OPT_S0_ON_MAIN 4096
#define DVGR_OPT_S0_ON_SIDE 8192
/* With the folded S1 backward the d p launch of a layer >= 1 has one reader, the query branch, and runs on the side stream
 * (round 6).  DVGR_OPT_DP_ON_MAIN keeps it on the caller's stream behind the dX contraction (A/B timing). */
#define DVGR_OPT_DP_ON_MAIN 16384
/* A fork right behind one launch lets its event ride on that launch's dispatch (hipExtLaunchKernelGGL stop event: no record packet
 * between the kernel and the next one of its stream; round 6).  DVGR_OPT_FORKS_BY_RECORD keeps hipEventRecord + wait everywhere:
 * A/B timing, and counter collection -- a dispatch that carries an event reads longer by the signal's path in per-kernel timings. */
#define DVGR_OPT_FORKS_BY_RECORD 32768

/* Scope-U loss evaluated by the forward itself (dvgr_opts.loss): the same arithmetic and outputs as
 * dvgr_scope_u_loss below, but the kernels are enqueued as soon as the last layer's list outputs exist and run on the
 * side stream under the forward's remaining contractions (last S4, MFB) instead of after them. */
typedef struct dvgr_loss_spec {
  float alpha, beta, dep_scale;
  float *loss3;                 /* device, 3 floats: {sum(vis), mean_l common, mean_l dependence}           */
  float *g_vis;                 /* (B, N, D), nullable: d loss / d vis (= 1); NULL also drops the sum(vis) term */
  float *g_com_app, *g_com_mot, *g_aq_fusion, *g_mq_fusion;   /* (L, B, N, D): d loss / d list tensor     */
} dvgr_loss_spec;

/* Per-call options of forward / backward (all fields may be zero). */
typedef struct dvgr_opts {
  int32_t train;       /* != 0: the 24 dropout sites per layer are active (p = 0.15, F.dropout semantics) */
  int32_t probe_site;  /* enum dvgr_probe_site, bracketed at layer `probe_layer`                           */
  int32_t probe_layer;
  int32_t flags;       /* DVGR_OPT_* bits                                                                  */
  uint64_t seed;       /* key of the counter-based dropout RNG; masks are recomputed, never stored         */
  void *probe_begin;   /* hipEvent_t, recorded before the probed launch (nullable)                         */
  void *probe_end;     /* hipEvent_t, recorded after it                                                    */
  void *const *layer_done_events; /* backward only, nullable: L+2 hipEvent_t, one per block of the flat gradient buffer
                          (dvgr_param_layer_range), recorded in the order the blocks become final: event[L] the MFB
                          block, event[L-1] .. event[0] the layer blocks (GAT + SF-attention parameters), event[L+1] the
                          query block (QueryAttn / QueryPunish parameters of all layers: their branch runs once, batched
                          over the layers, after the last layer) -- so a data-parallel caller can start the RCCL
                          all-reduce of a block while the rest of the backward is still running */
  const dvgr_loss_spec *loss;     /* forward only, nullable: also evaluate the scope-U loss and its gradients */
  void (*layer_done_cb)(int32_t index, void *user); /* backward only, nullable: called ON THE HOST, in the calling thread,
                          right after layer_done_events[index] has been enqueued -- the place to enqueue that slice's
                          all-reduce, so that the collectives enter their stream while the rest of the backward is still
                          being launched (after the call returns the device has usually finished the backward already) */
  void *layer_done_user;
  void *comm;          /* backward only, nullable: a dvgr_comm* (include/dvgr_comm.h).  The backward then enqueues the RCCL
                          all-reduce (mean over ranks, in place) of every block of `gparams` itself, on the communicator's
                          stream, behind the block's own event, the moment the block's last kernel has been enqueued -- no host
                          callback, no helper thread.  The caller makes its stream wait with dvgr_comm_join() before it reads
                          the gradients.  Independent of layer_done_events / layer_done_cb (which may stay NULL). */
  const void *p1t;     /* backward only, nullable: the transposed bf16 planes of every layer's project[0] weight as written
                          by dvgr_pack_p1t() from the CURRENT parameter planes.  The fused GAT-view backward (DVGR_OPT_NO_FUSED_
                          GAT_BWD) contracts against P1^T; without this pointer the call transposes the planes itself into its
                          workspace, one small launch per call (14 MB at C2) -- a caller whose parameters change once per
                          optimizer step prepares them once per step instead, next to the planes themselves. */
  const void *lo8;     /* forward and backward, nullable: the 8-bit block-scaled lo planes of the fused GAT-view kernels' weight
                          operands as written by dvgr_pack_lo8() from the CURRENT planes (Wcat for the forward, P1^T for the
                          backward).  Those launches then stream 3 bytes per weight element instead of 4 (csrc/dvgr_gatf.h, LO8);
                          without the pointer -- or with DVGR_OPT_NO_LO8 -- they stream the bf16 lo planes. */
} dvgr_opts;

/* Forward of the whole stacked unit: replaces DualVGRUnit_multiple.forward (model/models.py:
 * 121-173) and, inside it, QueryAttn/QueryPunish.forward (model/utils.py:66-105), punishGAT /
 * PunishGraphAttentionLayer.forward (model/GraphNN.py:95-178), AttentionSFGCN.forward
 * (model/Attention.py:20-23) and MFB.forward (model/fusions/fusions.py:419-453).
 * `params_lp` may be NULL for DVGR_F32; `opts` may be NULL (eval mode, no probe). */
int dvgr_unit_forward(const dvgr_dims *d, const float *params, const void *params_lp,
                      const dvgr_io *io, void *saved, size_t saved_bytes,
                      void *workspace, size_t workspace_bytes,
                      const dvgr_opts *opts, void *stream);

/* Backward of the same: replaces the autograd graph PyTorch builds for the modules above.
 * `opts->train` and `opts->seed` must equal the forward's. */
int dvgr_unit_backward(const dvgr_dims *d, const float *params, const void *params_lp,
                       const dvgr_io *io, const dvgr_grads *g,
                       const void *saved, size_t saved_bytes,
                       void *workspace, size_t workspace_bytes,
                       const dvgr_opts *opts, void *stream);

/* [first, last) element range of one block of the flat parameter / gradient buffer: layer < L: that layer's GAT and
 * SF-attention parameters; layer == L: the MFB block; layer == L + 1: the query block (QueryAttn and QueryPunish
 * parameters of all layers).  The blocks tile the buffer: [layer 0 | ... | layer L-1 | MFB | query]. */
int dvgr_param_layer_range(const dvgr_dims *d, int layer, int64_t *first, int64_t *last);

/* Scope-U scalar loss on the unit's outputs, value and gradient in one pass:
 *   loss = sum(vis) + alpha * mean_l common_loss(com_app_l, com_mot_l)
 *                   + beta * dep_scale * mean_l [loss_dependence(aq_fusion_l, com_app_l)
 *                                              + loss_dependence(mq_fusion_l, com_mot_l)]
 * common_loss / loss_dependence restate utils.py:10-31 of the reference; the weighting is train.py:148-154
 * (there the first term is the cross-entropy of the classifier, which is outside the unit).
 * `loss3` (device, 3 floats) receives {sum(vis), mean_l common, mean_l dependence}.  The outputs g_vis (B,N,D;
 * may be NULL) and g_com_app / g_com_mot / g_aq_fusion / g_mq_fusion (L,B,N,D) receive d loss / d tensor and can be
 * handed to dvgr_unit_backward as the upstream gradients.  Only io->vis and the four list tensors are read;
 * io->vis == NULL drops the first term (loss3[0] = 0, g_vis untouched): the auxiliary terms alone, for a caller whose
 * main loss is the classifier's cross-entropy.
 * `workspace` is the unit's workspace (dvgr_workspace_bytes). */
int dvgr_scope_u_loss(const dvgr_dims *d, const dvgr_io *io, float alpha, float beta, float dep_scale,
                      float *loss3, float *g_vis, float *g_com_app, float *g_com_mot,
                      float *g_aq_fusion, float *g_mq_fusion,
                      void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DVGR_H */
