/*
 * stereonet_hip.h — C ABI of libstereonet_hip.so, the MI355X (gfx950) replacement
 * for the Horizon BPU execution behind hobot_stereonet's StereonetNode.
 *
 * What this boundary replaces in the reference (paths under /root/reference):
 *   - hobot::dnn_node::DnnNode::Init()  -> model load     stereonet_infer/src/stereonet_node.cpp:44
 *   - DnnNode::GetModelInputSize / Model::Get*TensorProperties
 *                                                          stereonet_node.cpp:45,57-103
 *   - DnnNode::Run(inputs, output, is_sync, -1, -1)        stereonet_node.cpp:812 (async),
 *                                                          :177,:584,:968 (sync)
 *   - PreProcess::CvtNV12Data2Tensors (optional fused)     stereonet_infer/src/preprocess.cpp:913-1059
 *   - the NV12 side-by-side split in FeedImg (optional)    stereonet_node.cpp:705-738
 * The reference reaches all of these through the closed `dnn_node`/`libdnn`
 * (hbDNN*, hbSys*) API; hobot_stereonet_amd/csrc/compat/ re-implements exactly
 * the members the reference touches on top of the functions below (see
 * INTEGRATION.md for the binding a maintainer adds).
 *
 * Conventions: plain C types only; every function returns int, 0 = ok, <0 =
 * error (the reference's -1 convention, stereonet_infer/include/parser.h:37-39);
 * no exceptions cross the boundary.  Buffers are caller-owned.  `mem` says
 * whether data pointers are host (SN_MEM_HOST) or HIP device (SN_MEM_DEVICE)
 * memory.  `stream` is a hipStream_t passed as void* (NULL = the handle's own
 * stream, and the call returns after completion; non-NULL with SN_MEM_DEVICE =
 * work is only enqueued on that stream).  A handle is bound to one GPU; calls on
 * one handle must not overlap in time except sn_submit/sn_wait, which are
 * thread-safe (task_num requests in flight, stereonet_node.cpp:144).
 *
 * Tensor contract (unchanged from the reference):
 *   input  int8  NCHW [n][6][H][W]: L-Y, L-"U", L-"V", R-Y, R-"U", R-"V"; value = byte ^ 0x80
 *          (preprocess.cpp:999-1003,1033-1040)
 *   output int32 NCHW [n][1][H][W]: raw; disparity_px = raw * out_scale * 16 * 12 — the reference's literal
 *          factor (stereonet_node.cpp:282-288, parser.cpp:84-86, publisher_member_function.py:73-75) for every
 *          dmax, so the unmodified consumers recover pixels whatever D the model was built for
 *   optional float output [n][H][W]: disparity in px before int32 quantisation.
 */
#ifndef STEREONET_HIP_H_
#define STEREONET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sn_handle sn_handle;

/* Version of this binary interface.  3 = SN_PREC_AUTO exists and is what 0 ("default") selects, sn_io_info ends in
 * precision_selected, sn_get_refine_stats / sn_auto_* exist; 2 = the SN_PREC_* numbering below with 0 = SN_PREC_F16 and
 * exact fp32 = 3; version 1 (round 1) had 0 = exact fp32.  A caller built against an older header compares
 * SN_ABI_VERSION with sn_abi_version() at start-up instead of silently running in another arithmetic. */
#define SN_ABI_VERSION 3
int sn_abi_version(void);

enum {
  SN_OK = 0,
  SN_ERR_ARG = -1,       /* null / out-of-range argument, geometry mismatch          */
  SN_ERR_FILE = -2,      /* model_file missing or unreadable (stereonet_node.cpp:131) */
  SN_ERR_FORMAT = -3,    /* not an SNW1 weight file / wrong architecture header       */
  SN_ERR_DEVICE = -4,    /* HIP runtime error or no gfx950 device                     */
  SN_ERR_NOMEM = -5,
  SN_ERR_BUSY = -6,      /* sn_submit: no free task slot within the timeout           */
  SN_ERR_TICKET = -7     /* sn_wait: unknown or already-consumed ticket               */
};

enum { SN_MEM_HOST = 0, SN_MEM_DEVICE = 1 };

/* Arithmetic of the convolution contractions.  All variants accumulate in fp32. */
enum {
  SN_PREC_DEFAULT = 0,   /* a zero-initialised or NULL sn_config selects SN_PREC_AUTO             */
  SN_PREC_F16X3 = 1,     /* refinement tower on fp16 MFMA with hi/lo operand split (3 MFMAs per   */
                         /* product, ~2^-22 relative): fp32-class accuracy at 3/16 of the fp32    */
                         /* MFMA cost; activations stored as two fp16 tensors                     */
  SN_PREC_F16 = 2,       /* refinement tower on plain fp16 MFMA operands (3x3 weights rounded per  */
                         /* kernel so that every kernel's tap sum survives: no coherent offset);  */
                         /* low-resolution branch on 22-bit split fp16 operands.  Its error is     */
                         /* proportional to what the refinement adds to the map: over 8 weight    */
                         /* draws (profiles/r05_epe_sensitivity_*.txt) EPE vs the fp32 oracle is    */
                         /* 1.8e-4 .. 8.3e-4 px per pixel of mean |D r|, i.e. < 1e-3 px is only    */
                         /* GUARANTEED up to ~1.1 px of mean residual at 1280x720 (typically up   */
                         /* to ~2 px); a hierarchical model holds at head gain 1 only.  Forcing    */
                         /* this mode is the caller's statement that the model is inside that.    */
  SN_PREC_FP32 = 3,      /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere                   */
  SN_PREC_AUTO = 4       /* the default: SN_PREC_F16 while the model stays inside the fp16 tower's */
                         /* envelope, SN_PREC_F16X3 otherwise.  Every head kernel sums |D r| (the  */
                         /* refinement statistic, sn_get_refine_stats); the first call of a handle */
                         /* also runs one pair in both arithmetics and measures their distance.    */
                         /* A call whose statistic (or self-check) predicts EPE > 1e-3 px is       */
                         /* REPEATED in SN_PREC_F16X3 before it returns, and the handle stays      */
                         /* there until the statistic has been back inside 0.8 of the envelope    */
                         /* for 8 calls.  Calls that only enqueue work (device buffers + a caller */
                         /* stream) cannot be repeated: the first such call of a handle blocks    */
                         /* for the self-check, later ones act on the statistic of the call        */
                         /* before (sn_auto_* below is the state machine, pure functions).         */
};

typedef struct sn_config {
  int device;        /* HIP device ordinal, -1 = current device                                 */
  int max_batch;     /* largest n for sn_infer_batch; <=0 -> 1                                   */
  int width;         /* 0 = take from the model file header                                      */
  int height;        /* 0 = take from the model file header                                      */
  int dmax;          /* max disparity D (multiple of 16, <= 256); 0 = from the model file        */
  int precision;     /* SN_PREC_*; 0 = SN_PREC_AUTO                                               */
  int task_num;      /* async slots for sn_submit; <=0 -> 4 (stereonet_node.cpp:144)              */
  int refine_chunk;  /* pairs per refinement-tower launch; <=0 -> sized by work per launch, about    */
                     /* 5.5 Mpx (6 at 1280x720, 8 at 1248x384, max 8); the per-layer forms           */
                     /* (SN_FUSE=0, SN_PREC_F16X3, SN_PREC_FP32) keep the Infinity-Cache sizing       */
  int piece;         /* pairs per low-resolution piece of the pipeline; <=0 -> 16 (SN_PREC_FP32: the    */
                     /* first piece of a call is 2-4 pairs, nothing overlaps its low-res branch)     */
} sn_config;

typedef struct sn_io_info {
  int width, height, dmax;
  int in_channels;        /* 6 */
  int max_batch, precision, task_num, device;
  float out_scale;        /* 2.60443857769133e-6 (stereonet_node.cpp:282) */
  size_t in_bytes;        /* per pair: 6*H*W  (int8)  */
  size_t out_bytes;       /* per pair: 4*H*W  (int32) */
  double flops_per_pair;  /* algorithmic conv FLOPs (2*MAC), DESIGN.md §5 */
  int refine_chunk;       /* pairs per refinement-tower launch actually in use */
  int piece;              /* pairs per low-resolution piece actually in use    */
  int tower_streams;      /* tower chunks in flight (1 or 2)                   */
  int refine_levels;      /* 1 = single-scale refinement; 4 = hierarchical (towers at 1/8, 1/4, 1/2, 1), as the
                           * model file says (weights.py: header word 72) */
  int precision_selected; /* arithmetic the NEXT call runs in: = precision unless that is SN_PREC_AUTO
                           * (then SN_PREC_F16 or SN_PREC_F16X3) */
} sn_io_info;

/* DnnNode::Init + Model introspection ------------------------------------------------------- */
int sn_create(const char *model_file, const sn_config *cfg, sn_handle **out);
int sn_destroy(sn_handle *h);
int sn_get_io_info(const sn_handle *h, sn_io_info *info);
const char *sn_strerror(int code);
const char *sn_last_error(const sn_handle *h);   /* detail of the last failure on this handle; h = NULL: of the
                                                  * last failed sn_create on the calling thread */

/* Refinement statistic and SN_PREC_AUTO --------------------------------------------------------------------------
 * The reference loads an opaque model_file and only checks that it exists (stereonet_node.cpp:131-136); whether the
 * fp16 tower keeps north_star's 1e-3 px on it depends on how far its refinement moves the map.  Every head kernel
 * therefore accumulates the sum of |D_k r_k| over the pixels it writes (level k of the refinement, in level-k pixels;
 * one 64-bit fixed-point atomic per wave, outputs bit-unchanged), all precision modes. */
typedef struct sn_refine_stats {
  int levels;                       /* refinement levels of the model (1 or 4)                                        */
  int precision;                    /* as configured (SN_PREC_*)                                                      */
  int precision_selected;           /* arithmetic the next call runs in (differs from `precision` under SN_PREC_AUTO)  */
  int precision_last;               /* arithmetic the most recent call's maps were computed in                        */
  uint64_t calls, pairs;            /* completed calls / stereo pairs since sn_create (every entry point)             */
  uint64_t switches;                /* SN_PREC_AUTO: changes of arithmetic                                            */
  uint64_t reruns;                  /* SN_PREC_AUTO: calls repeated in SN_PREC_F16X3 before they returned              */
  double level_px[4];               /* last call: mean |D_k r_k| of level k (0 = full resolution), level-k pixels      */
  double residual_px;               /* last call: sum_k 2^k level_px[k] = full-resolution pixels the refinement adds   */
  double running_px;                /* exponential mean of residual_px over the calls (weight 1/4)                    */
  double envelope_px;               /* SN_PREC_F16 is trusted while residual_px stays below this (shape class)         */
  double limit_px;                  /* the threshold in force (sn_auto_limit_px)                                      */
  double selfcheck_epe_px;          /* mean |F16 - F16X3| of the self-check pair, < 0 = not measured                   */
  double selfcheck_residual_px;     /* residual_px of that pair                                                       */
} sn_refine_stats;
int sn_get_refine_stats(sn_handle *h, sn_refine_stats *out);

/* SN_PREC_AUTO's decision as pure functions (no GPU; tests/test_auto_precision.py).  sn_auto_observe folds the statistic of
 * one finished call in and returns the arithmetic of the next one; a caller that ran the call in SN_PREC_F16 and gets
 * SN_PREC_F16X3 back repeats the call.
 *   limit: envelope_px (the shape class's envelope = the budget over the WORST error-per-pixel of the measured weight
 *          draws) until the self-check has measured THIS model's error per pixel of residual (epe_per_px); then
 *          min(SN_AUTO_ENVELOPE_CAP * envelope_px, SN_AUTO_BUDGET_PX / epe_per_px).
 *   F16 -> F16X3: residual > limit (at once).  F16X3 -> F16: SN_AUTO_CALM_CALLS consecutive calls with residual <
 *   SN_AUTO_REENTRY * limit. */
#define SN_AUTO_BUDGET_PX 0.85e-3   /* of north_star's 1e-3 px: the rest is SN_PREC_F16X3's own distance to the oracle */
#define SN_AUTO_ENVELOPE_CAP 4.0    /* a measured slope may widen the class envelope by at most this factor            */
#define SN_AUTO_REENTRY 0.8
#define SN_AUTO_CALM_CALLS 8
typedef struct sn_auto_state {
  int mode;             /* SN_PREC_F16 or SN_PREC_F16X3: arithmetic of the next call                              */
  int calm;             /* consecutive SN_PREC_F16X3 calls whose residual sat inside the re-entry band            */
  double envelope_px;   /* measured envelope of the shape class (sn_auto_envelope_px)                             */
  double epe_per_px;    /* self-check: mean |F16 - F16X3| per pixel of residual; 0 = not measured                 */
  double running_px;    /* exponential mean of the residual, < 0 before the first call                            */
  uint64_t switches;
} sn_auto_state;
double sn_auto_envelope_px(int refine_levels);          /* 1: single-scale; 4: hierarchical                        */
int sn_auto_init(sn_auto_state *s, int refine_levels);
double sn_auto_limit_px(const sn_auto_state *s);
int sn_auto_observe(sn_auto_state *s, double residual_px);   /* returns the new s->mode                            */

/* DnnNode::Run, synchronous form (stereonet_node.cpp:968) ----------------------------------- */
/* out_i32 and out_disp may each be NULL (but not both). */
int sn_infer_i8(sn_handle *h, const int8_t *in_nchw6, int32_t *out_i32, float *out_disp,
                int mem, void *stream);
int sn_infer_batch(sn_handle *h, int n, const int8_t *in_nchw6, int32_t *out_i32, float *out_disp,
                   int mem, void *stream);

/* PreProcess::CvtNV12Data2Tensors on the GPU (preprocess.cpp:913-1059), bit-exact, including the
 * reference's planar-I420 reading of the NV12 chroma (preprocess.h:131-133). */
int sn_preprocess_nv12(sn_handle *h, const uint8_t *left_nv12, const uint8_t *right_nv12,
                       int w, int h_px, int8_t *out_nchw6, int mem, void *stream);
/* FeedImg's split (stereonet_node.cpp:705-738) + CvtNV12Data2Tensors + Run in one call: takes the
 * raw 2W x H side-by-side NV12 message payload.  out_tensor (nullable) receives the int8 model
 * input the reference would have built. */
int sn_infer_sbs_nv12(sn_handle *h, const uint8_t *sbs_nv12, int w2, int h_px, int32_t *out_i32,
                      float *out_disp, int8_t *out_tensor, int mem, void *stream);

/* The same split + mapping for n side-by-side frames (n * 3*H*W bytes -> n * 6*H*W bytes), no inference: the batched
 * ingest of a streaming host (bench.py --stream) that ships camera frames instead of model tensors. */
int sn_preprocess_sbs_nv12_batch(sn_handle *h, int n, const uint8_t *sbs_nv12, int w2, int h_px, int8_t *out_nchw6,
                                 int mem, void *stream);

/* DnnNode::Run, asynchronous form (stereonet_node.cpp:812): host buffers only.  sn_submit copies
 * the input and returns at once with a ticket; up to task_num tickets are in flight;
 * timeout_ms < 0 waits for a free slot forever (the reference passes -1).  sn_wait blocks until the
 * ticket's pair is done, fills the host outputs given at submit, and reports the device time. */
int sn_submit(sn_handle *h, const int8_t *in_nchw6_host, int32_t *out_i32_host, float *out_disp_host,
              int timeout_ms, uint64_t *ticket);
/* The same with FeedImg's raw 2W x H side-by-side NV12 frame as the input (stereonet_node.cpp:705-738 + preprocess.cpp:
 * 913-1059 run on the GPU): half the host-to-device bytes of sn_submit. */
int sn_submit_nv12(sn_handle *h, const uint8_t *sbs_nv12_host, int w2, int h_px, int32_t *out_i32_host,
                   float *out_disp_host, int timeout_ms, uint64_t *ticket);
int sn_wait(sn_handle *h, uint64_t ticket, float *infer_ms);
int sn_synchronize(sn_handle *h);

/* Multi-GPU form: the independent pairs of one batch sharded over the GPUs of one node -------------------------
 * The reference keeps task_num = 4 independent frames in flight behind one Run() call site
 * (stereonet_node.cpp:144,812); this spreads such units of work over devices instead: contiguous shards (the first
 * n % ndev shards get one extra pair), one host thread + one engine per GPU, weights replicated, NO data-path
 * collective; the single exchange is the gather of the maps to the root.  cfg->device is ignored, cfg->max_batch is
 * the largest TOTAL n; devices == NULL selects 0..ndev-1.
 *   sn_mgpu_infer_batch         host buffers: every device copies its shard in and its maps out — the host is the root.
 *   sn_mgpu_infer_batch_device  in_per_device[k] = shard k resident on device k; maps gathered in batch order into
 *                               out_* on device 0 over xGMI: one grouped RCCL ncclSend/ncclRecv exchange per batch
 *                               (the default when ndev > 1 and librccl loads), or hipMemcpyPeerAsync from each peer over
 *                               its own link (fallback; SN_MGPU_GATHER=peer forces it).  = submit + wait below.
 *   sn_mgpu_submit_device /     the asynchronous form of the same (the reference's async Run with task slots): returns a
 *   sn_mgpu_wait                ticket once every device has its shard enqueued; SN_MGPU_SLOTS = 2 tickets may be in
 *                               flight (per-device staging is double buffered), so the gather of batch k overlaps the
 *                               compute of batch k + 1.  Inputs and root buffers belong to the call until the wait.
 * Results are bit-identical to sn_infer_batch on one GPU (same kernels, no cross-pair reduction). */
typedef struct sn_mgpu sn_mgpu;
int sn_mgpu_shard(int n, int ndev, int k, int *first, int *count);         /* pure shard arithmetic */
int sn_mgpu_create(const char *model_file, const sn_config *cfg, const int *devices, int ndev, sn_mgpu **out);
int sn_mgpu_destroy(sn_mgpu *m);
int sn_mgpu_get_info(const sn_mgpu *m, int *ndev, int *per_device_batch, int *gather_kind /* 1 peer copy, 2 RCCL */);
int sn_mgpu_get_handle(sn_mgpu *m, int k, sn_handle **h);                  /* the engine of shard k (borrowed) */
/* The borrowed engine shares its workspace with the sn_mgpu_* calls: use it (sn_infer_*, sn_submit, ...) only while
 * no sn_mgpu_submit_device ticket is in flight and no other sn_mgpu_* call runs; sn_mgpu_infer_batch itself returns
 * SN_ERR_BUSY while tickets are outstanding. */
int sn_mgpu_infer_batch(sn_mgpu *m, int n, const int8_t *in_nchw6_host, int32_t *out_i32_host, float *out_disp_host);
int sn_mgpu_infer_batch_device(sn_mgpu *m, int n, const int8_t *const *in_per_device, int32_t *out_i32_root,
                               float *out_disp_root);
int sn_mgpu_submit_device(sn_mgpu *m, int n, const int8_t *const *in_per_device, int32_t *out_i32_root,
                          float *out_disp_root, uint64_t *ticket);      /* SN_ERR_BUSY: two tickets in flight */
int sn_mgpu_wait(sn_mgpu *m, uint64_t ticket);                            /* SN_ERR_TICKET: unknown / consumed   */
const char *sn_mgpu_last_error(const sn_mgpu *m);
/* The ticket / buffer-slot bookkeeping of the asynchronous form as pure functions (no GPU; tests): tickets count from
 * 1, ticket t uses slot t % SN_MGPU_SLOTS, a slot is busy from submit to the wait of its ticket. */
#define SN_MGPU_SLOTS 2
typedef struct sn_mgpu_ring {
  uint64_t next;
  uint64_t slot_ticket[SN_MGPU_SLOTS];
} sn_mgpu_ring;
int sn_mgpu_ring_init(sn_mgpu_ring *r);
int sn_mgpu_ring_submit(sn_mgpu_ring *r, uint64_t *ticket, int *slot);
int sn_mgpu_ring_wait(sn_mgpu_ring *r, uint64_t ticket, int *slot);

/* Measurement hooks (bench.py): per-stage device time of the most recent sn_infer_batch, taken
 * with hipEvents on the stream the kernels ran on.  Stage ids: */
enum {
  SN_STAGE_FEATURES = 0,   /* Siamese tower, both eyes            */
  SN_STAGE_AGGREGATE = 1,  /* cost volume + 3-D convs + soft-argmin */
  SN_STAGE_REFINE = 2,     /* upsample + refinement tower + output epilogue */
  SN_STAGE_REFINE_CONV = 3,/* the span of the 12 C->C 3x3 convs inside REFINE (first chunk) */
  SN_STAGE_TOTAL = 4,
  SN_STAGE_DOMINANT = 5,   /* the launches sn_get_dominant_kernel describes, first refinement chunk, timed one by one
                            * (the streamed residual blocks); equals REFINE_CONV when the tower runs layer by layer */
  SN_STAGE_COUNT = 6
};
int sn_set_profiling(sn_handle *h, int enable);
int sn_get_stage_ms(sn_handle *h, float *ms, int count);
/* launches of the dominant kernel in the last call and its algorithmic FLOPs / HBM bytes per launch */
int sn_get_dominant_kernel(sn_handle *h, char *name, size_t name_cap, int *launches,
                           double *flops_per_launch, double *bytes_per_launch);

/* Parity hooks (tests only; they run the product kernels on caller data, host memory) -------- */
/* one C->32 convolution through the MFMA kernel: in [cin][h][w] fp32, wt [32][cin][k][k], out [32][ho][wo].
 * lrelu bit 0 = LeakyReLU, bit 1 = use the split-operand fp16 kernel of the fp16 modes (cin 32; 3x3 stride 1 or
 * 5x5 stride 2), bit 2 (with bit 1) = run it on split-slot tensors (hi/lo fp16, the fp16 modes' low-resolution
 * activation format) through the weights-stationary kernel; the hook converts to and from that layout; bit 3 = the
 * fp32 tower kernel k_ref_conv_f32 (cin 32, 3x3, w % 4 == 0) instead of the generic fp32 kernel; bit 4 (with bits 1
 * and 2) = the kernel of the zero-bordered tensors: k_down_x3s_dma (5x5 stride 2, no residual) or k_feat_x3s_dma (3x3,
 * residual allowed: added in place as the feature tower does). */
int sn_dbg_conv2d(sn_handle *h, const float *in, int cin, int h_px, int w, const float *wt,
                  const float *bias, int k, int stride, int dil, int lrelu, const float *residual,
                  float *out);
/* the first down-conv (3->32, 5x5, stride 2, no activation) of both eyes through the fp16-MFMA kernel of the fp16
 * modes: in6 int8 [6][h][w] (model input), wt [32][3][5][5], out [2][32][ho][wo] with ho/wo = ceil16(h|w)/2;
 * tc = 32 or 64 selects the tile width */
int sn_dbg_down0(sn_handle *h, const int8_t *in6, int h_px, int w, const float *wt, const float *bias, int tc,
                 float *out);
/* The rounding SN_PREC_F16 applies to the 3x3 weights of its refinement towers at model load (host only, no device):
 * w [nkernels][9] fp32 -> out [nkernels][9], every value one of the two fp16 numbers enclosing its input, chosen per kernel
 * so that the SUM of the nine rounding errors is smallest (csrc/stereonet_hip.hip round_kernel_sum_preserving;
 * SN_W_ROUND=rne restores round-to-nearest in the engine). */
int sn_dbg_round_kernels_f16(const float *w, int nkernels, float *out);
/* The first TWO down-convs (3->32 and 32->32, both 5x5 stride 2, no activation between them) folded into one 13x13
 * stride-4 convolution, as the fp16 modes run them (csrc/sn_down01.hpp; SN_DOWN01=0 restores the two kernels).
 * sn_dbg_compose_down01 is the host-side fold alone (no device): w0 [32][3][5][5], b0 [32], w1 [32][32][5][5], b1 [32]
 * -> weff [9][32][3][13][13], beff [9][32]; class = 3 * row class + column class, each {first, inner, last} row / column
 * of the quarter-resolution map (down-conv 1's zero padding of the half-resolution map drops taps there).
 * sn_dbg_down01 runs both eyes through the pipeline's two kernels: in6 int8 [6][h][w] -> out [2][32][ho][wo] with
 * ho/wo = ceil16(h|w)/4. */
int sn_dbg_compose_down01(const float *w0, const float *b0, const float *w1, const float *b1, float *weff,
                          float *beff);
int sn_dbg_down01(sn_handle *h, const int8_t *in6, int h_px, int w, const float *w0, const float *b0,
                  const float *w1, const float *b1, float *out);
/* the refinement input conv (4->32, 3x3, LeakyReLU) through the fp16-MFMA kernel of the fp16 modes:
 * disp_low fp32 [hp/16][wp/16] (full-resolution px / 16 units as the soft-argmin head writes it), in6 int8 [6][h][w],
 * wt [32][4][3][3]; out fp32 [32][hp][wp] (hp/wp = ceil16) read back from the fp16 NCHW8c tensor(s); split != 0
 * selects the hi/lo output of SN_PREC_F16X3 */
int sn_dbg_refin(sn_handle *h, const float *disp_low, const int8_t *in6, int h_px, int w, int dmax,
                 const float *wt, const float *bias, int split, float *out);
/* one 3x3x3 32->32 conv3d (+bias, optional LeakyReLU): in [32][d][h][w] -> out [32][d][h][w]; lrelu bits as above,
 * bit 3 (with bits 1 and 2): the aggregation kernel of the zero-bordered volumes (k_agg_x3s_dma) */
int sn_dbg_conv3d(sn_handle *h, const float *in, int d, int h_px, int w, const float *wt,
                  const float *bias, int lrelu, float *out);
/* one 32->32 3x3 conv (dilation 1/2/4/8) through the fp16 refinement-tower kernel: in / residual / out are
 * fp32 [32][h][w] on the host; the hook converts to the kernel's fp16 NCHW8c layout and back.
 * lrelu bit 0 = LeakyReLU; bit 1 / bit 2 force the 8x64 / 8x32 tile variant of the dilation-1 / -2 kernel (default: the
 * width the engine picks for the launch from its tile count). */
int sn_dbg_ref_conv_f16(sn_handle *h, const float *in, int h_px, int w, const float *wt, const float *bias,
                        int dil, int lrelu, const float *residual, float *out);
/* the same layer through the split-operand (SN_PREC_F16X3) kernel */
int sn_dbg_ref_conv_f16x3(sn_handle *h, const float *in, int h_px, int w, const float *wt, const float *bias,
                          int dil, int lrelu, const float *residual, float *out);
/* one residual block y = lrelu(x + conv2(lrelu(conv1(x)+b1)) + b2) of the fp16 tower; fp32 [32][h][w] host tensors.
 * dil = 1 / 2 / 4 / 8 in bits 0..7; bits 8.. select the form: 0 = two convolution launches, 2 = the row-streaming fused
 * kernel the pipeline runs by default (every dilation). */
int sn_dbg_ref_block_f16(sn_handle *h, const float *in, int h_px, int w, const float *w1, const float *b1,
                         const float *w2, const float *b2, int dil, float *out);
/* the same block on split operands (SN_PREC_F16X3): form 0 = two k_ref_conv_f16x3 launches, 1 = the row-streaming fused kernel
 * (sn_stream_block_x3.hpp) — bit-identical to form 0 */
int sn_dbg_ref_block_f16x3(sn_handle *h, const float *in, int h_px, int w, const float *w1, const float *b1,
                           const float *w2, const float *b2, int dil, int form, float *out);
/* The LAST residual block of the fp16 tower followed by the refinement head (conv 3x3 32 -> 1, disp = relu(up + D r), wire
 * quantisation) on n images: fp32 host tensors in [n][32][hk][wk] (the level's padded activation, rounded to fp16 by the hook),
 * low [n][hk / ups][wk / ups] (the map the level starts from; ups = 16: soft-argmin map, 2: the level below), head_w [32][9].
 * form 0 = streamed block + k_head_final_f16 (two launches), 1 = the tail form the pipeline runs (one launch, y never
 * written); out_disp / out_raw [n][h_out][w_out], h_out <= hk, w_out <= wk.  The two forms must agree bit for bit. */
int sn_dbg_ref_tail_f16(sn_handle *h, int n, const float *in, int hk, int wk, const float *w1, const float *b1,
                        const float *w2, const float *b2, const float *head_w, float head_b, const float *low, int ups,
                        float dnorm, int h_out, int w_out, int form, float *out_disp, int32_t *out_raw);
/* intermediates of the most recent batch-1 inference: "feat_l" / "feat_r" [32][hl][wl],
 * "cost" [Dl][hl][wl], "disp_low" [hl][wl], and for a hierarchical model "level1" .. "level3" (the map of that
 * refinement level, [Hp/2^k][Wp/2^k]); returns the element count in *n (dst may be NULL to query). */
int sn_dbg_read(sn_handle *h, const char *what, float *dst, size_t cap, size_t *n);
/* Parse()'s dequantisation + depth (stereonet_infer/src/parser.cpp:84-86) on the GPU, for n maps of the model's size:
 *     dis = (float)raw * out_scale;   depth_m = (float)((double)(focal_px * baseline_mm) / (dis * 16.0 * 12.0) / 1000.0)
 * with the reference's float / double mix, so the result is bit-identical to the host Parse (raw = 0 gives IEEE inf).
 * The reference's constants are focal_px = 527.1931762695312, baseline_mm = 119.89382172 (parser.cpp:70-71).
 * raw / depth_m: host or device buffers per `mem`; disp_px (nullable) receives dis * 16 * 12 as Parse's disparity. */
int sn_depth_from_raw(sn_handle *h, int n, const int32_t *raw, float focal_px, float baseline_mm, float *depth_m,
                      float *disp_px, int mem, void *stream);

/* Measurement hook (bench.py --emulate-root-ingress): a device-to-device copy of `bytes` bytes by a kernel of exactly
 * `workgroups` workgroups of 256 threads on `stream` — the footprint of one RCCL receive (a few channels = a few
 * workgroups per peer), so that the tax of the gather root's ingress on a concurrently running batch can be measured
 * on one GPU.  dst / src: device pointers, 16-byte aligned; bytes a multiple of 16. */
int sn_dbg_copy_limited(void *dst, const void *src, size_t bytes, int workgroups, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STEREONET_HIP_H_ */
