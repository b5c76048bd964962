/*
 * wavernn_amd.h -- C ABI of the MI355X-native WaveRNN generate path (libwavernn_amd.so).
 *
 * Drop-in boundary for ONE hot path of fatchord/WaveRNN: the per-sample loop of
 * `WaveRNN.generate()` (reference models/fatchord_version.py:169-264).  The reference has no FFI of its
 * own (it is pure Python on PyTorch); these entry points are what a binding for that path binds, one per
 * reference stage, and each cites the reference lines it replaces.  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every `const float*` marked "device" is a HIP device pointer owned by the caller
 *     (e.g. a torch tensor's data_ptr()); the library BORROWS it for the duration of the call.
 *   - all functions return WRNN_OK (0) or a negative error code; wrnn_last_error() returns a message for the
 *     calling thread's last failure.  Nothing here synchronises the device except the functions that say so.
 *   - a wrnn_pack is immutable after creation: any number of threads / streams may generate from one pack at the same
 *     time, each with its own workspace (and timer).  Nothing is read from the environment.
 *   - work is enqueued on the hipStream_t passed as `void* stream` (NULL = default stream).
 *   - arithmetic type: float32 (same as the reference).  No CPU fallback exists: without a HIP device every
 *     compute entry point fails with WRNN_ERR_NO_DEVICE.
 */
#ifndef WAVERNN_AMD_H
#define WAVERNN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WRNN_ABI_VERSION 9   /* v9 (round 6): wrnn_pack_sparse_fc_blocks -- block-sparse Linear layers in wrnn_sparse_kernel.  v8 (round 6): WRNN_ALGO_OCTO (wrnn_octo_kernel: one 512-thread workgroup per CU, matrix waves + service waves; dense MOL).  v7 (round 5): WRNN_ALGO_CHAIN (wrnn_chain_kernel, what `auto` runs for <= 128 segments of a dense model, MOL or 9-bit RAW); WRNN_ALGO_SPARSE is the rebuilt wrnn_sparse_kernel (slabbed, resumable, takes mel_stage) and what `auto` picks for a qualifying pack; wrnn_options.depth does not apply to it.  v6 (round 4): wrnn_options.mel_stage & co + wrnn_pre_upsample_rows -- the last up-sampling stage formed inside wrnn_duo_kernel.  v5 (round 4): WRNN_ALGO_DUO runs RAW too; `auto` never degrades inside the library (WRNN_ERR_RESIDENCY: the caller re-plans); tuning bits per kernel */

enum {
    WRNN_OK = 0,
    WRNN_ERR_ARG = -1,        /* bad argument (shape, NULL pointer, unsupported dims) */
    WRNN_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime failure */
    WRNN_ERR_HIP = -3,        /* a HIP API call failed; see wrnn_last_error() */
    WRNN_ERR_WORKSPACE = -4,  /* workspace too small */
    WRNN_ERR_KERNEL = -5,     /* a kernel reported a failure (bounded spin expired, bad state) */
    WRNN_ERR_RESIDENCY = -6   /* persistent grid cannot be co-resident on this device */
};

enum { WRNN_MODE_RAW = 0, WRNN_MODE_MOL = 1 };  /* reference: WaveRNN(mode='RAW'|'MOL'), fatchord_version.py:97-104 */

/* Loop kernel selection. */
enum {
    WRNN_ALGO_AUTO = 0,     /* shipped dims: WRNN_ALGO_SPARSE when the pack qualifies (wrnn_pack_sparse_blocks() > 0) on a 256-CU device, else
                               WRNN_ALGO_CHAIN (MOL or RAW with 512 classes, <= 128 segments, 256 CUs, dense pack), else WRNN_ALGO_DUO (MOL, or RAW
                               with 512 classes; >= 128 CUs),
                               else WRNN_ALGO_STREAM; other dims: wrnn_generic_kernel */
    WRNN_ALGO_STREAM = 1,   /* one workgroup per folded segment, weights streamed from L2/MALL each step: the generic fallback
                               (any class count, any device size) and the on-GPU cross-check */
    WRNN_ALGO_LOOP = 2,     /* role-split pipelined persistent kernel (RAW and MOL): up to 4 clusters of 64 CUs, each with a full
                               fp32 copy of the loop weights in registers, up to 8 groups of <= 16 segments in flight per cluster,
                               tag-free activation exchange in MFMA-fragment order (csrc/wrnn_loop.hip) */
    WRNN_ALGO_DUO = 6,      /* the loop kernel cut for TWO workgroups per CU (MOL and RAW): four roles -- rnn1 / rnn2 x {W_ih + fc, W_hh [+ fc3 and
                               sampling]} -- of <= 128 weight registers, 128 workgroups per 64-CU cluster, so that one wave's MFMAs overlap
                               the other's loads, pointwise math and barrier waits (csrc/wrnn_duo.hip).  What `auto` runs for a dense pack
                               beyond 128 segments */
    WRNN_ALGO_CHAIN = 7,    /* the single-stream latency kernel (MOL and 9-bit RAW, <= 256 segments, 256 CUs): one workgroup per CU, <= 4 groups of <= 16 segments per
                               64-CU cluster, both halves of a GRU cell's rows in one workgroup (gh never leaves the registers), rnn2 + fc1 + fc2
                               on one XCD (csrc/wrnn_chain.hip).  What `auto` runs for one utterance (<= 128 segments) of a dense model */
    WRNN_ALGO_OCTO = 8,     /* the WAVE-SPECIALISED form of the dense loop kernel (MOL, 256 CUs; round 6): ONE 512-thread workgroup per CU -- four matrix
                               waves (W_ih and W_hh of the CU's 16 units in registers, the fc tile in LDS: operand by LDS-DMA, MFMA block, partial tiles)
                               and four service waves (partial sums, GRU cell, publishes, conditioning, fc3 + sampling) that meet through LDS counters;
                               gh never leaves the CU.  Same split, workspace, exchange buffer and state layout as WRNN_ALGO_DUO (csrc/wrnn_octo.hip) */
    WRNN_ALGO_SPARSE = 5    /* block-sparse GRU kernel (MOL; BASELINE config 5): needs GRU matrices whose 16x1 block rows keep <= 64 columns
                               (wrnn_pack_sparse_blocks) and a 256-CU device; fc1 / fc2 are gathered too when they are block-sparse (wrnn_pack_sparse_fc_blocks); 16 clusters of 16 CUs, ONE group of <= 16 segments each: a
                               step is the latency of one chain, sixteen chains run side by side (csrc/wrnn_sparse.hip) */
};

/*
 * Host-side view of the loop weights, exactly the tensors `WaveRNN.generate()` touches inside its loop
 * (state_dict keys in comments; shapes for rnn_dims=H, fc_dims=F, feat_dims=M, aux_dims=A=res_out_dims/4,
 * n_classes=C).  fatchord_version.py:115-123 (definition), :208-223 (use), :273-279 (GRU cell views).
 * All pointers are HOST pointers to contiguous row-major float32.
 */
typedef struct wrnn_weights {
    int32_t rnn_dims;    /* H: 512 for the MFMA kernels (loop / duo / sparse / stream); any other value <= 2048 runs on wrnn_generic_kernel */
    int32_t fc_dims;     /* F: 512 for the MFMA kernels; <= 2048 otherwise */
    int32_t feat_dims;   /* M: 80 for the MFMA kernels; feat + aux <= 1024 otherwise */
    int32_t aux_dims;    /* A: 32 for the MFMA kernels */
    int32_t n_classes;   /* C: 30 (MOL) or 2**bits (RAW); shipped dims: RAW 2..512 classes (the loop kernel needs 512, others stream) */
    int32_t mode;        /* WRNN_MODE_* */
    const float *I_w, *I_b;                          /* I.weight (H,1+M+A), I.bias (H) */
    const float *w_ih1, *w_hh1, *b_ih1, *b_hh1;      /* rnn1.weight_ih_l0 (3H,H), weight_hh_l0 (3H,H), biases (3H) */
    const float *w_ih2, *w_hh2, *b_ih2, *b_hh2;      /* rnn2.weight_ih_l0 (3H,H+A), weight_hh_l0 (3H,H), biases (3H) */
    const float *fc1_w, *fc1_b;                      /* fc1.weight (F,H+A), fc1.bias (F) */
    const float *fc2_w, *fc2_b;                      /* fc2.weight (F,F+A), fc2.bias (F) */
    const float *fc3_w, *fc3_b;                      /* fc3.weight (C,F), fc3.bias (C) */
} wrnn_weights;

/* Opaque device-resident weight pack (replaces the per-call `get_gru_cell` views, fatchord_version.py:178-179). */
typedef struct wrnn_pack wrnn_pack;

/*
 * Geometry of one batched (folded) or unbatched generation -- the arguments of
 * `fold_with_overlap(x, target, overlap)` (fatchord_version.py:293-340) without materialising the fold:
 * segment b, step t reads the un-folded conditioning at position p = b*stride + t; positions p >= L are the
 * reference's zero padding (:326-330).  Unbatched: B=1, T=L, stride=0.
 */
typedef struct wrnn_geometry {
    int32_t B;        /* number of folded segments (num_folds) */
    int32_t T;        /* steps per segment: target + 2*overlap (batched) or L (unbatched) */
    int32_t stride;   /* target + overlap */
    int32_t L;        /* un-folded conditioning length in samples (N*hop) */
    int32_t hop;      /* samples per mel frame (aux is constant over a frame: Stretch2d, :57-61,:84) */
    int32_t n_frames; /* rows of `aux` (= L / hop) */
} wrnn_geometry;

/* HIP-event timer owned by the CALLER (one per in-flight call): wrnn_generate* records an event pair around every loop-kernel
 * launch it enqueues.  Keeps the weight pack immutable, so one pack can serve several streams / threads at once. */
typedef struct wrnn_timer wrnn_timer;

/* What a wrnn_generate* call decided (filled synchronously, before the call returns). */
typedef struct wrnn_run_info {
    const char *kernel;      /* "wrnn_duo_kernel" / "wrnn_chain_kernel" / "wrnn_sparse_kernel" / "wrnn_loop_kernel" / "wrnn_stream_kernel" / "wrnn_generic_kernel" */
    int32_t units_per_wg;    /* hidden units per workgroup (16; sparse: 64; stream: 0) */
    int32_t clusters;        /* independent CU clusters, each holding one copy of the weights */
    int32_t depth;           /* groups of <= 16 segments in flight per cluster */
    int32_t rounds;          /* rounds of clusters x depth groups the segments were cut into */
    int32_t slab_steps;      /* steps per launch of a persistent kernel = one conditioning slab (wrnn_loop_kernel: steps of hoisted cI
                                materialised at a time; duo / sparse: steps one set of per-segment aux tables covers).  wrnn_plan_segments()
                                does not know the hop: there this is an UPPER BOUND -- a duo / sparse call caps it at 6 hops + 1 steps */
    int32_t launches;        /* loop-kernel launches enqueued */
} wrnn_run_info;

/*
 * Per-call options (replaces the environment variables ABI v2 read at call time).  Zero-initialise, set struct_bytes =
 * sizeof(wrnn_options); NULL options = all defaults.  Everything here is read during the call only.
 */
typedef struct wrnn_options {
    int32_t struct_bytes;
    int32_t algo;            /* WRNN_ALGO_*                                                                  (default AUTO) */
    int32_t depth;           /* groups in flight per cluster, 1..8 (sparse: always 1); 0 = the library picks */
    int32_t clusters;        /* loop kernel: clusters to use (1, 2 or 4); 0 = the library picks */
    int32_t cond_valu;       /* stream kernel: 1 = hoisted conditioning on VALU instead of MFMA (cross-check) */
    int32_t slab_steps;      /* loop kernel: conditioning slab length in steps; 0 = sized to ~96 MB */
    int32_t t_begin, t_end;  /* run steps [t_begin, t_end) of the T; 0,0 = all.  t_begin > 0 CONTINUES the call that ended at
                                t_begin on the same workspace (wrnn_loop / _duo / _sparse / _chain_kernel); `noise` then covers [t_begin, t_end) only,
                                `out` / force_x / logits always the whole [.., T] tensors */
    int32_t tuning;          /* A/B switches for measurements, PER KERNEL (0 = the measured defaults; results never depend on them):
                                wrnn_loop_kernel: bit 0 = no one-stage look-ahead of the exchange loads, bit 1 = full __syncthreads() fences
                                  at the stage barriers, bit 2 = no fused stages, bit 3 = RAW sampled by role A alone, bit 4 = RAW: one
                                  sampling workgroup per slot, bit 5 / bit 6 = never / always start a stage with the pending back half
                                  (default: up to 2 groups in flight), bit 7 = library exp / tanh in the MoL gate math;
                                wrnn_duo_kernel: bit 0 = the ih workgroups load a stage's operand into registers BEHIND the pending back half, bit 9 = in front
                                  of it (default: from 4 groups in flight), bit 1 = they fetch it into LDS one stage ahead (on request only), bit 2 = re-fill
                                  the exchange ring with the sentinel before EVERY launch, bit 3 = no second request for x_{t-1}, bit 4 = a fresh y2 request
                                  at the top of the sampling stage, bit 5 = sampling stage at wave priority 0, bit 7 = the residual input word requested beside
                                  the operand loads (round 5), bit 20 = the hh workgroups' gh block as ONE unbroken MFMA stream (bits 21-25: yield length /
                                  spacing of the broken one), bit 6 = placement read-out through phase_clocks (test hook), bit 8 = every layer written
                                  through (no XCD-local plain stores), bit 14 = with phase_clocks: the stage time line of three steps as well
                                  (phase_clocks then holds [256 * 32 + 512 * 512] words; scripts/gpu_duo_trace.py);
                                wrnn_sparse_kernel, wrnn_chain_kernel: bits 2, 8 as wrnn_duo_kernel; wrnn_sparse_kernel: bit 11 = the DENSE fc stages on a pack whose
                                  Linear layers are block-sparse too; wrnn_chain_kernel: bit 5 = MoL's fc3 tiles read from LDS (round 5).
                                When the two-workgroups-per-CU grid of wrnn_duo_kernel is refused the call returns WRNN_ERR_RESIDENCY; the
                                caller may run it again with WRNN_ALGO_LOOP (another workspace layout: query its size) or _STREAM. */
    const float *force_x;    /* test hook, device [n,T]: value fed back as x_t instead of the sample (teacher forcing) */
    float *logits;           /* test hook, device [T,n,C]: fc3 output of every step (:223) */
    unsigned long long *phase_clocks; /* profiling hook, device [256 workgroups][32] zeroed by the caller: wrnn_loop_kernel (MOL and RAW) and
                                wrnn_duo_kernel (MOL) add their per-(phase, stage segment) shader clocks (layouts: csrc/wrnn_loop.hip,
                                csrc/wrnn_duo.hip, "PROF"); with tuning bit 6: the placement read-out of wrnn_duo / _sparse_kernel instead */
    wrnn_timer *timer;       /* optional: time the loop kernel(s) of this call */
    wrnn_run_info *info;     /* optional out */
    /* optional progress read-out (ABI v4) -- the reference's `gen_display` (fatchord_version.py:241, :267-271: a rate line every
     * 100 steps of its Python loop).  Here the loop is one persistent kernel per conditioning slab, so the read-out exists per
     * slab: after the launches of every slab a host function is enqueued on `stream` (hipLaunchHostFunc) that calls
     * progress(steps_done, T, n_segments, progress_user) from a HIP runtime thread once the device has really got there.  No
     * synchronisation, nothing inside the kernel.  The callback must not call into HIP.  Loop kernel only (the other kernels are
     * one launch: they report once, at the end). */
    void (*progress)(int32_t steps_done, int32_t T, int32_t n_segments, void *user);
    void *progress_user;
    /* the LAST up-sampling stage inside the loop (ABI v6; wrnn_duo_kernel, wrnn_sparse_kernel, wrnn_chain_kernel -- any other kernel: WRNN_ERR_ARG).  The reference
     * up-samples the mel in three Stretch2d + box-conv stages and crops `indent` samples off both ends (fatchord_version.py:73-80,
     * :86-88) before the loop reads one [M] row per sample.  With mel_stage = 1 the `mels_up` argument of wrnn_generate* is the INPUT of
     * the last of those stages instead -- [mel_rows][M], what wrnn_pre_upsample_rows() writes: 1 / mel_scale of the rows -- and the loop
     * forms the row of step t of segment b from the three input rows its 2 * mel_scale + 1 taps reach, at the un-cropped position
     * j = seg_pos[b] + t + seg_moff[b] (one utterance: seg_moff = indent = pad * hop; utterance u of a concatenation whose S2 blocks
     * follow one another: indent * (2 u + 1)).  seg_pos / seg_lim / L / aux keep their meaning (the cropped time line).
     * This build: mel_scale == 11. */
    int32_t mel_stage;       /* 0 = `mels_up` is the up-sampled mel [L, M] (default) */
    int32_t mel_rows;        /* rows of `mels_up` when mel_stage = 1 */
    int32_t mel_scale;       /* stretch factor of the last stage */
    const float *mel_taps;   /* HOST [2 * mel_scale + 1]: upsample.up_layers.5.weight */
    const int32_t *seg_moff; /* HOST [n_segments] */
} wrnn_options;

const char *wrnn_last_error(void);
int wrnn_abi_version(void);

/* Number of compute units of HIP device `device` (or <0). */
int wrnn_device_cus(int device);

/* Upload + re-layout the loop weights for `device`.  Synchronous (uses its own stream and waits). */
int wrnn_pack_create(const wrnn_weights *w, int device, wrnn_pack **out);
void wrnn_pack_destroy(wrnn_pack *p);
/* bytes of loop weights the reference streams per step (the W of SURVEY.md section 8d) */
size_t wrnn_pack_weight_bytes(const wrnn_pack *p);
/* block sparsity of the pack's GRU matrices: the largest number of non-zero 16x1 blocks in any (matrix, gate, 16-row) block
 * row -- positive when WRNN_ALGO_SPARSE can run this pack (<= 64, MOL), negated when it cannot */
int wrnn_pack_sparse_blocks(const wrnn_pack *p);
/* (v9) the same figure for fc1 / fc2 (their first H columns; 32 block rows of 16 each): positive when the Linear layers are block-sparse
 * too -- the reference's pruning recipe prunes them with the GRUs, notebooks/"Pruning - Scratchpad.ipynb":199-204 -- and
 * wrnn_sparse_kernel runs its GATHERED fc stages (no dense fc1 / fc2 MFMA blocks on the chain); negated: dense fc stages */
int wrnn_pack_sparse_fc_blocks(const wrnn_pack *p);

/* Workspace (device bytes) wrnn_generate needs for this geometry under these options (NULL = defaults).  With the loop
 * kernel it does not grow with T: conditioning is produced in slabs (942 segments x 12,100 steps: < 1 GB). */
size_t wrnn_workspace_bytes(const wrnn_pack *p, const wrnn_geometry *g, const wrnn_options *opt);

/*
 * The loop: replaces fatchord_version.py:192-245 (state init, the `for i in range(seq_len)` body incl.
 * sample_from_discretized_mix_logistic / softmax+Categorical, and the stack/transpose gather).
 *
 *   mels_up  device [L, M]          un-folded up-sampled mel  (`mels` of :186 before the fold)
 *   aux      device [n_frames, 4A]  MelResNet output per FRAME (`aux` of :186 before Stretch2d repeats it)
 *   noise    device, the sampling noise in the order the reference draws it:
 *              MOL: [T, 11*B]: per step 10*B uniforms (segment-major, mixture-minor; distribution.py:106)
 *                   followed by B uniforms (distribution.py:118), all U(1e-5, 1-1e-5)
 *              RAW: [T, B, C] Exp(1) variates (Categorical.sample -> multinomial)
 *   out      device [B, T]          the (B,T) tensor of :243 (samples in [-1,1]; RAW: 2*idx/(C-1)-1)
 *
 * Enqueues on `stream`; does not synchronise.  Call wrnn_status() after synchronising to learn whether
 * the kernels completed (bounded spins never hang: they give up and report).
 */
int wrnn_generate(const wrnn_pack *p, const wrnn_geometry *g, const float *mels_up, const float *aux,
                  const float *noise, float *out, void *workspace, size_t workspace_bytes, const wrnn_options *opt,
                  void *stream);

/*
 * The same loop over an explicit SEGMENT TABLE: segment b, step t reads conditioning position
 * p = seg_pos[b] + t; positions p >= seg_lim[b] are zero padding.  This is `fold_with_overlap`
 * (fatchord_version.py:293-340) applied to SEVERAL utterances whose un-folded conditioning was concatenated
 * (every utterance must start on a frame boundary: offset % hop == 0): utterance u at sample offset o_u with
 * L_u samples contributes segments with seg_pos = o_u + i*(target+overlap), seg_lim = o_u + L_u.  It is what
 * lets one launch serve a whole corpus shard (BASELINE config 4) -- the reference calls generate() once per
 * utterance (gen_wavernn.py:26-35).  seg_pos / seg_lim are HOST arrays of n_segments int32.
 *   mels_up device [L, M], aux device [n_frames, 4A] (concatenated), noise/out as in wrnn_generate with
 *   B = n_segments.
 */
int wrnn_generate_segments(const wrnn_pack *p, int32_t n_segments, int32_t T, const int32_t *seg_pos,
                           const int32_t *seg_lim, int32_t L, int32_t hop, int32_t n_frames,
                           const float *mels_up, const float *aux, const float *noise, float *out,
                           void *workspace, size_t workspace_bytes, const wrnn_options *opt, void *stream);
size_t wrnn_workspace_bytes_segments(const wrnn_pack *p, int32_t n_segments, int32_t T, int32_t n_frames,
                                     const wrnn_options *opt);

/* What wrnn_generate_segments WOULD run for this many segments under these options (kernel, split, rounds, slab length),
 * without launching anything.  `launches` is left 0. */
int wrnn_plan_segments(const wrnn_pack *p, int32_t n_segments, int32_t T, const wrnn_options *opt, wrnn_run_info *out);

/* Synchronises `stream`, reads the kernel status words from `workspace`.  WRNN_OK or WRNN_ERR_KERNEL. */
int wrnn_status(void *workspace, void *stream);

/* Test hook: one exchanged activation layer (0 h1, 1 h2, 2 y1, 3 y2, 4 RAW logits) of (cluster, slot) at ring position `ring`
 * (= step % 4) of the loop kernel's exchange buffer, un-permuted from MFMA-fragment order into host_out[16 segments][512].
 * Call after a run with the same (n_segments, T, n_frames, opt).  Synchronises the device. */
int wrnn_debug_read_exchange(const wrnn_pack *p, void *workspace, int32_t n_segments, int32_t T, int32_t n_frames,
                             const wrnn_options *opt, int cluster, int slot, int layer, int ring, float *host_out);

/* Timer objects (see wrnn_options.timer).  wrnn_timer_ms synchronises on the recorded events and returns the SUM of the
 * loop-kernel launch durations of the last call that used the timer, including the calls that continued it with
 * t_begin > 0 (<0 if none); wrnn_timer_launches their count. */
int wrnn_timer_create(int device, wrnn_timer **out);
void wrnn_timer_destroy(wrnn_timer *t);
float wrnn_timer_ms(wrnn_timer *t);
int wrnn_timer_launches(const wrnn_timer *t);

/*
 * Pre-loop stage: `UpsampleNetwork.forward` (fatchord_version.py:82-89) = MelResNet (:31-48) + ResBlocks (:13-28) on
 * f32 MFMA, and the Stretch2d / box-filter up-sampling of the mel (:51-61, :73-80, :86-88).  Host pointers, float32,
 * row-major, in the reference's state-dict order (keys in comments, C = compute_dims, R = res_out_dims).
 */
typedef struct wrnn_pre_weights {
    int32_t feat_dims;            /* 80 */
    int32_t compute_dims;         /* C (shipped: 128) */
    int32_t res_out_dims;         /* R (shipped: 128) */
    int32_t res_blocks;           /* number of ResBlocks */
    int32_t pad;                  /* 2.  Any dims the reference's constructor takes (fatchord_version.py:64-71): (80, 128, 128, pad 2) run the f32-MFMA
                                     MelResNet kernel, everything else wrnn_resnet_generic_kernel (round 5) */
    int32_t upsample_factors[3];  /* (5, 5, 11) */
    const float *conv_in_w;       /* upsample.resnet.conv_in.weight (C, feat, 2*pad+1) */
    const float *bn_in;           /* upsample.resnet.batch_norm.{weight,bias,running_mean,running_var}: [4][C] */
    const float *res_w;           /* upsample.resnet.layers.{i}.{conv1,conv2}.weight: [res_blocks][2][C][C] */
    const float *res_bn;          /* upsample.resnet.layers.{i}.batch_norm{1,2}.{weight,bias,running_mean,running_var}: [res_blocks][2][4][C] */
    const float *conv_out_w;      /* upsample.resnet.conv_out.weight (R, C) */
    const float *conv_out_b;      /* upsample.resnet.conv_out.bias (R) */
    const float *up_w;            /* upsample.up_layers.{1,3,5}.weight, concatenated: (2*s0+1) + (2*s1+1) + (2*s2+1) taps */
} wrnn_pre_weights;
typedef struct wrnn_pre wrnn_pre;

int wrnn_pre_create(const wrnn_pre_weights *w, int device, wrnn_pre **out);
void wrnn_pre_destroy(wrnn_pre *p);
int wrnn_pre_hop(const wrnn_pre *p);                                   /* product of the upsample factors */
size_t wrnn_pre_workspace_bytes(const wrnn_pre *p, int32_t n_frames);
/* mel: device [feat][n_frames] (the (1, feat, N) tensor generate() receives, fatchord_version.py:169,183);
 * mels_up: device [n_frames*hop][feat]; aux: device [n_frames][R] (per FRAME: the x hop Stretch2d repeat of :83-85 is
 * left to the loop).  Several utterances: call once per utterance with offset output pointers.  Asynchronous on `stream`. */
int wrnn_pre_upsample(const wrnn_pre *p, const float *mel, int32_t n_frames, float *mels_up, float *aux,
                      void *workspace, size_t workspace_bytes, void *stream);
/* The same without the last Stretch2d + conv stage and its crop (ABI v6): mel_rows: device [(n_frames + 2 pad) * s0 * s1][feat], the input
 * of the last stage, which wrnn_duo_kernel then forms inside the loop (wrnn_options.mel_stage = 1); aux as above. */
int wrnn_pre_upsample_rows(const wrnn_pre *p, const float *mel, int32_t n_frames, float *mel_rows, float *aux,
                           void *workspace, size_t workspace_bytes, void *stream);
const char *wrnn_pre_last_error(void);

/*
 * Post-loop stage on the device, float64 like the reference: `.astype(np.float64)` (:245), `decode_mu_law` (utils/dsp.py:
 * 98-103; `lut` = the expansion of every class value, computed by the caller with the reference's numpy expression, or NULL
 * for MOL / mu_law off), `xfade_and_unfold` (:342-405; `fade_in` / `fade_out` = its two overlap-long float64 ramps) and the
 * truncation + tail fade of :255-258 (`tail` = linspace(1, 0, 20*hop)).  Utterance u owns segments
 * [first[u], first[u]+folds[u]) of `segments` [n][T] and writes out[out_off[u] .. out_off[u+1]) (= wave_len_u samples).
 * All pointers are DEVICE pointers.  batched = 0: one segment per utterance, no cross-fade.  Asynchronous on `stream`.
 */
int wrnn_post_unfold(const float *segments, int32_t T, int32_t n_utt, const int32_t *first, const int32_t *folds,
                     const int64_t *out_off, const double *lut, int32_t n_classes, const double *fade_in,
                     const double *fade_out, int32_t overlap, const double *tail, int32_t tail_len, int32_t batched,
                     double *out, void *stream);
const char *wrnn_post_last_error(void);

/*
 * BASELINE config 3's caller: the Tacotron decoder loop as one persistent kernel (on hardware since round 3:
 * tests/test_gpu_config3.py compares it with the reference's own output, tests/golden/tacotron_decoder_200f.npz).  Replaces the per-frame loop of `Tacotron.generate()` (reference models/tacotron.py:396-414) around
 * `Decoder.forward` (:218-279) for one sentence; the encoder (:24-39, once per sentence) and the post-net stay with the caller.
 * All pointers are DEVICE pointers to contiguous float32 tensors in the reference's state-dict layouts (`decoder.*`).
 */
typedef struct wrnn_taco_weights {
    uint32_t struct_bytes;
    int32_t n_mels, prenet1, prenet2, decoder_dims, encoder_width, lstm_dims, attn_filters, attn_kernel;   /* 80 256 128 256 256 512 32 31 */
    const float *prenet_fc1_w, *prenet_fc1_b, *prenet_fc2_w, *prenet_fc2_b;               /* [256][80] [256] [128][256] [128] */
    const float *attn_rnn_w_ih, *attn_rnn_w_hh, *attn_rnn_b_ih, *attn_rnn_b_hh;           /* GRUCell [768][384] [768][256] [768] [768] */
    const float *attn_W_w, *attn_W_b, *attn_conv_w, *attn_L_w, *attn_L_b, *attn_v_w;      /* LSA: [256][256] [256] [32][2][31] [256][32] [256] [256] */
    const float *rnn_input_w, *rnn_input_b;                                               /* [512][512] [512] */
    const float *rnn1_w_ih, *rnn1_w_hh, *rnn1_b_ih, *rnn1_b_hh;                           /* LSTMCell [2048][512] x2, [2048] x2 */
    const float *rnn2_w_ih, *rnn2_w_hh, *rnn2_b_ih, *rnn2_b_hh;
    const float *mel_proj_w;                                                              /* [n_mels * max_r][512] */
} wrnn_taco_weights;

typedef struct wrnn_taco_call {
    uint32_t struct_bytes;
    int32_t n;               /* encoder positions (<= 1024) */
    int32_t r, max_r;        /* frames per decoder step (decoder.r) / of the mel_proj view (:262) */
    int32_t max_steps;       /* decoder steps at most (the reference's `steps` / r) */
    float stop_threshold;    /* :411 */
    const float *seq;        /* [n][256] encoder_seq (:403) */
    const float *seq_proj;   /* [n][256] encoder_seq_proj (:404) */
    float *mel_out;          /* [max_steps][n_mels][r]: frame block of every step */
    float *scores_out;       /* [max_steps][n]: attention of every step (:413) */
    int32_t *steps_done;     /* decoder steps run, including the one that met the stop test */
    void *workspace;         /* wrnn_taco_workspace_bytes() */
    size_t workspace_bytes;
    void *stream;
    int32_t variant;         /* 0 = auto; 1 = the flag-barrier kernel (weights re-read from L2 every step, any CU count);
                                2 = the register-resident kernel (128 co-resident workgroups, tagged exchange, r <= 8);
                                3 = 2 with per-layer shader clocks of workgroup 0 in the last 24 x 8 bytes of the workspace */
} wrnn_taco_call;

size_t wrnn_taco_workspace_bytes(void);
/* Asynchronous on `stream`.  WRNN_ERR_RESIDENCY if the cooperative grid is refused. */
int wrnn_taco_decode(int device, const wrnn_taco_weights *w, const wrnn_taco_call *c);
/* Synchronises `stream`; out4 = {failure flag, code, workgroup, barrier} of the decode that used `workspace` (all 0 = clean). */
int wrnn_taco_status(const void *workspace, unsigned *out4, void *stream);
const char *wrnn_taco_last_error(void);

/* The bidirectional GRU that ends a CBHG (reference models/tacotron.py:95, applied at :137: `x, _ = self.rnn(x)`), one sequence,
 * hidden size 128 (encoder and post-net of the reference's hparams): one persistent workgroup per direction with W_hh in
 * registers.  gi_* = W_ih x + b_ih for all frames (a plain GEMM: the caller's).  Errors: wrnn_taco_last_error(). */
typedef struct wrnn_bigru_call {
    uint32_t struct_bytes;
    int32_t T, hidden;                       /* frames; 128 */
    const float *gi_fwd, *gi_rev;            /* [T][3 hidden] (gates r, z, n as in nn.GRU) */
    const float *w_hh_fwd, *w_hh_rev;        /* [3 hidden][hidden] */
    const float *b_hh_fwd, *b_hh_rev;        /* [3 hidden] */
    float *out;                              /* [T][2 hidden]: forward | reverse, as nn.GRU(bidirectional=True) returns */
    void *stream;
} wrnn_bigru_call;
int wrnn_bigru(int device, const wrnn_bigru_call *c);

/* Self tests of the device primitives (MFMA fragment layout, inter-workgroup granule all-gather).
 * Synchronous.  WRNN_OK or an error with a message. */
int wrnn_selftest(int device, int which);
/* microseconds per all-gather round measured by the last wrnn_selftest(device, 2) (<0 if none) */
float wrnn_selftest_metric(void);

#ifdef __cplusplus
}
#endif
#endif /* WAVERNN_AMD_H */
