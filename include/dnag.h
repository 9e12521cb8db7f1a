/*
 * dnag.h -- C ABI of the MI355X-native reverse-diffusion sampling hot path.
 *
 * Every entry point takes plain pointers and sizes.  Device pointers are HIP device
 * addresses on the CURRENT device; `stream` is a hipStream_t passed as void*.  The caller
 * owns every buffer; the library allocates no device memory.  All entry points return 0 on
 * success and a negative code on error (DNAG_E_*); dnag_last_error() gives the message of
 * the calling thread's last failure.  Nothing here synchronises the host with the device
 * unless the comment says so.
 *
 * Streams: work is ordered on the `stream` passed in.  dnag_pc_sample (guided) and
 * dnag_cate_dsm_step also use one stream of the library's own per device (created, with two
 * events, by the first such call on the device: make it outside a stream capture), forked
 * from and joined to `stream` by events inside the call; results do not depend on it.
 * DNAG_OVERLAP=0 in the environment (read at every call) keeps everything on `stream`.
 *
 * Reference citations are relative to /root/reference/NAS-Bench-201/.
 */
#ifndef DNAG_H
#define DNAG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNAG_E_ARG      (-1)   /* bad argument / unsupported shape */
#define DNAG_E_WORKSPACE (-2)  /* workspace too small */
#define DNAG_E_HIP      (-3)   /* a HIP call failed (message has hipGetErrorString) */

/* Score network shape (CATE, models/cate.py:108-143; values: configs/tr_scorenet.py:60-101). */
typedef struct dnag_cate_dims {
    int32_t max_node;      /* N: 8 (NAS-Bench-201) or 20 (OFA) */
    int32_t n_vocab;       /* V: 7 or 9 */
    int32_t d_model;       /* 64 */
    int32_t n_head;        /* 8 */
    int32_t d_ff;          /* 128 */
    int32_t n_layers;      /* 12 */
    int32_t stagewise_pos; /* 0: PositionalEncoding_Cell, 1: _StageWise (cate.py:133-136) */
} dnag_cate_dims;

/* Surrogate shape (MetaNeuralPredictor, models/digcn_meta.py:56-107). */
typedef struct dnag_pred_dims {
    int32_t max_node;      /* 8 */
    int32_t n_vocab;       /* graph_encoder.initial_hidden = 7 */
    int32_t gcn_hidden;    /* 144 */
    int32_t gcn_layers;    /* 4 */
    int32_t linear_hidden; /* 128 */
    int32_t hs;            /* 32 (512 in the pickled Transfer-NAG config) */
    int32_t nz;            /* 56 */
} dnag_pred_dims;

/* OFA surrogate shape (MetaPredictorCATE, models/cate.py:251-305; values:
 * MobileNetV3/configs/tr_meta_surrogate_ofa.py:112-145). */
typedef struct dnag_predcate_dims {
    int32_t max_node;      /* 20 */
    int32_t n_vocab;       /* 9 */
    int32_t d_model;       /* 64 */
    int32_t n_head;        /* 2 */
    int32_t d_ff;          /* 32 */
    int32_t n_layers;      /* 2 */
    int32_t hs;            /* 512 */
    int32_t nz;            /* 56 */
} dnag_predcate_dims;

int         dnag_abi_version(void);
const char *dnag_last_error(void);

/* ---- host-side weight packing (no device access) -------------------------------------------
 * `state_flat` is the concatenation of the state_dict tensors in state_dict() order, fp32
 * (models/cate.py:110-126, models/transformer.py:69-179).  `tables` are
 * the two non-parameter constants the reference builds in its constructors:
 *   CATE:      pos-enc table [max_node][d_model] (cate.py:211-248) then timestep frequencies
 *              [d_model/2] (cate.py:363-368);
 *   predictor: timestep frequencies [gcn_hidden/2] (digcn_meta.py:162-168).
 * `packed` receives the kernel layout (LayerNorm affine folded into the following linear,
 * DiGCN weights also stored transposed).  Sizes are in floats. */
size_t dnag_cate_state_floats(const dnag_cate_dims *d);
size_t dnag_cate_table_floats(const dnag_cate_dims *d);
size_t dnag_cate_packed_floats(const dnag_cate_dims *d);
int    dnag_cate_pack(const dnag_cate_dims *d, const float *state_flat, const float *tables, float *packed);

/* Only the parameters used while sampling are consumed, in this order: gcn.l.weight1,
 * gcn.l.weight2 (l = 0..L-1), fc1.weight, timeEmb1.{weight,bias}, timeEmb2.{weight,bias},
 * pred_fc.0.{weight,bias}, pred_fc.2.{weight,bias} (the set-encoder stays in PyTorch and runs
 * once per call, digcn_meta.py:150-156). */
size_t dnag_pred_state_floats(const dnag_pred_dims *d);
size_t dnag_pred_table_floats(const dnag_pred_dims *d);
size_t dnag_pred_packed_floats(const dnag_pred_dims *d);
int    dnag_pred_pack(const dnag_pred_dims *d, const float *state_flat, const float *tables, float *packed);

/* MetaPredictorCATE: the parameters used while sampling, in state_dict order: opEmb.fc, timeEmb1,
 * timeEmb2, graph_encoder.* (layers then norms), final.linears.{0,1,2}, regeress.linears.{0,1},
 * pred_fc.0, pred_fc.2 (weight then bias each).  `tables` = timestep frequencies [d_model/2]. */
size_t dnag_predcate_state_floats(const dnag_predcate_dims *d);
size_t dnag_predcate_table_floats(const dnag_predcate_dims *d);
size_t dnag_predcate_packed_floats(const dnag_predcate_dims *d);
int    dnag_predcate_pack(const dnag_predcate_dims *d, const float *state_flat, const float *tables, float *packed);
size_t dnag_predcate_workspace_bytes(const dnag_predcate_dims *d, int batch);

/* Bytes of device workspace needed for a batch of `batch` graphs (`pd` may be NULL). */
size_t dnag_workspace_bytes(const dnag_cate_dims *cd, const dnag_pred_dims *pd, int batch, int n_steps);
/* as above for guided sampling with the OFA surrogate */
size_t dnag_workspace_bytes_predcate(const dnag_cate_dims *cd, const dnag_predcate_dims *pd, int batch, int n_steps);

/* ---- score network forward ------------------------------------------------------------------
 * Replaces CATE.forward(X, time_cond, maskX) (models/cate.py:145-164, eval mode).
 * x [batch][N][V]; time_cond [batch] if time_stride==1, [1] if 0; mask [batch][N][N] if
 * mask_stride==1, [1][N][N] if 0 (query i sees key j iff mask[i][j]!=0, transformer.py:16);
 * out [batch][N][V].  `concat_out` (may be NULL) receives the all-layer concat [batch*N][L*d]
 * (cate.py:83-86), `attn0_out` (may be NULL) layer-0 attention probabilities [batch][H][N][N]. */
int dnag_cate_forward(const dnag_cate_dims *d, const float *packed, const float *x,
                      const float *time_cond, int time_stride, const float *mask, int mask_stride,
                      float *out, float *concat_out, float *attn0_out, int batch,
                      void *workspace, size_t workspace_bytes, void *stream);

/* ---- surrogate forward + guidance gradient ---------------------------------------------------
 * Replaces MetaNeuralPredictor.forward with cached D_mu (digcn_meta.py:109-159) and
 * get_classifier_grad_fn (models/utils.py:126-147, regress=True): y [batch] (may be NULL) and
 * grad [batch][N][V] = d(sum_b y_b)/dx (may be NULL; caller applies the 'min' sign).
 * d_mu [nz] is the dataset embedding (digcn_meta.py:131-139) computed once per call. */
size_t dnag_pred_workspace_bytes(const dnag_pred_dims *d, int batch);   /* batch = 1 when time_stride == 0 */
int dnag_pred_forward_grad(const dnag_pred_dims *d, const float *packed, const float *x,
                           const float *time_cond, int time_stride, const float *mask, int mask_stride,
                           const float *d_mu, float *y, float *grad, int batch,
                           void *workspace, size_t workspace_bytes, void *stream);

/* dnag_pred_forward_grad for callers that cannot allocate or bring a workspace (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results as
 * dnag_pred_forward_grad minus the workspace: the time-embedding staging ([batch][144] floats) lives in a device
 * array inside the library (2.25 MiB per GPU, held until the process ends).  batch > DNAG_NOALLOC_MAX_BATCH is
 * rejected with DNAG_E_ARG.  The call allocates nothing (no hipMalloc / hipFree, no torch).  Every call on one GPU
 * shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls on one stream do not. */
#define DNAG_NOALLOC_MAX_BATCH 4096
int dnag_pred_forward_grad_noalloc(const dnag_pred_dims *d, const float *packed, const float *x,
                                   const float *time_cond, int time_stride, const float *mask, int mask_stride,
                                   const float *d_mu, float *y, float *grad, int batch, void *stream);

/* Same for the OFA surrogate: replaces MetaPredictorCATE.forward with cached D_mu
 * (models/cate.py:311-360) and get_classifier_grad_fn (models/utils.py:126-147). */
int dnag_predcate_forward_grad(const dnag_predcate_dims *d, const float *packed, const float *x,
                               const float *time_cond, int time_stride, const float *mask, int mask_stride,
                               const float *d_mu, float *y, float *grad, int batch,
                               void *workspace, size_t workspace_bytes, void *stream);

/* ---- predictor-corrector sampler -------------------------------------------------------------
 * Replaces the loops of pc_sampler (sampling.py:424-445) and pc_conditional_sampler
 * (sampling.py:534-577): predictors ReverseDiffusionPredictor (sampling.py:203-213),
 * EulerMaruyamaPredictor (:189-200), NonePredictor (:216-224); correctors LangevinCorrector
 * (:227-256), NoneCorrector (:259-267); on a VPSDE (sde_lib.py:145-211) or a VESDE (:253-300),
 * continuous=True, probability_flow=False.
 *
 * `sched` is a HOST array [n_steps][DNAG_SCHED_COLS] built by the host mirrors of VPSDE / VESDE
 * (diffusionnag_amd/sde_lib.py) with the reference's own fp32 arithmetic.  Columns, VPSDE first: */
#define DNAG_SCHED_COLS 8
#define DNAG_SCHED_LABEL 0  /* time_cond: 999 t_i (models/utils.py:93) | sigma(t_i) (:109)                        */
#define DNAG_SCHED_STD   1  /* marginal_prob std at t_i (sde_lib.py:179-188) | unused (score is not rescaled, :116) */
#define DNAG_SCHED_ALPHA 2  /* alphas[k_i] (sampling.py:238-240) | 1 (:242)                                        */
#define DNAG_SCHED_SQRTA 3  /* reverse_diffusion: sqrt(alphas[k_i]) (sde_lib.py:203-207) | 1 (f = 0, :298)
                             * euler_maruyama:    -0.5 beta(t_i) (sde_lib.py:169-173) | 0 (:274)                   */
#define DNAG_SCHED_G     4  /* reverse_diffusion: G = sqrt(discrete_betas[k_i]) (sde_lib.py:204) |
                             *                        sqrt(sigma_k^2 - sigma_{k-1}^2) (:299)
                             * euler_maruyama:    diffusion * sqrt(-dt) (sampling.py:199)                          */
#define DNAG_SCHED_G2    5  /* reverse_diffusion: G * G in fp32 (sde_lib.py:122); euler_maruyama: diffusion ** 2 (:100) */
#define DNAG_SCHED_T     6  /* t_i                                                                                 */
#define DNAG_SCHED_DT    7  /* euler_maruyama: dt = -1/N (sampling.py:195)                                         */

#define DNAG_PREDICTOR_NONE              0
#define DNAG_PREDICTOR_REVERSE_DIFFUSION 1
#define DNAG_PREDICTOR_EULER_MARUYAMA    2
#define DNAG_SCORE_VP 0     /* score = -model / std  (models/utils.py:103) */
#define DNAG_SCORE_VE 1     /* score = model         (models/utils.py:116-117) */

typedef int (*dnag_langevin_exchange_fn)(void *user, void *stream);   /* see dnag_pc_args.langevin_exchange */

typedef struct dnag_pc_args {
    /* networks */
    const dnag_cate_dims *cate_dims;
    const float *cate_packed;          /* device */
    const dnag_pred_dims *pred_dims;   /* NULL: unconditional */
    const float *pred_packed;          /* device, or NULL */
    const dnag_predcate_dims *predcate_dims;   /* OFA surrogate instead of pred_dims (at most one of the two) */
    const float *predcate_packed;      /* device, or NULL */
    const float *d_mu;                 /* device [nz], or NULL */
    float classifier_scale;            /* sampling.py:505 */
    float label_sign;                  /* +1 for labels='max', -1 for 'min' (models/utils.py:136-139) */
    /* problem */
    int32_t batch;
    const float *mask;                 /* device [N][N] (mask[0] of the caller's tensor, sampling.py:438) */
    const float *sched;                /* HOST [n_steps][DNAG_SCHED_COLS] */
    int32_t n_steps;                   /* number of PC steps to run (sde.N for a full trajectory) */
    float snr;                         /* sampling.py:251 */
    int32_t corrector;                 /* 1: LangevinCorrector, 0: NoneCorrector */
    int32_t predictor;                 /* DNAG_PREDICTOR_* */
    int32_t score_mode;                /* DNAG_SCORE_VP or DNAG_SCORE_VE */
    float prior_scale;                 /* x0 = prior_scale * N(0,1) when x0 == NULL: 1 (sde_lib.py:191) | sigma_max (:285) */
    /* noise: tape mode if noise_corr/noise_pred != NULL ([n_steps][batch][N][V] each, device,
     * the order of randn_like at sampling.py:247 and :210); otherwise on-device Philox4x32-10 +
     * Box-Muller keyed by (seed, stream_offset + element index, step). */
    const float *x0;                   /* device [batch][N][V] prior sample (sampling.py:436) or NULL (Philox) */
    const float *noise_corr;
    const float *noise_pred;
    uint64_t seed;
    uint64_t stream_offset;            /* first global element index of this shard (multi-GPU) */
    /* outputs (device) */
    float *x_out;                      /* [batch][N][V] inverse_scaler(x_mean_last) = (x_mean+1)/2 (sampling.py:445) */
    float *x_last;                     /* optional: noisy x after the last predictor step, raw scale */
    float *x_mean_last;                /* optional: x_mean of the last predictor step, raw scale */
    /* optional per-step records for parity tests (device, may be NULL): [n_steps][batch][N][V] */
    float *rec_score_c, *rec_total_c, *rec_x_corr, *rec_score_p, *rec_total_p, *rec_x_mean, *rec_x_pred;
    float *rec_cgrad_c, *rec_cgrad_p;
    float *rec_step_size;              /* [n_steps][3]: grad_norm, noise_norm, step_size */
    /* workspace */
    void *workspace;
    size_t workspace_bytes;
    void *stream;
    /* guidance weighting options of the MobileNetV3 tree's sampler (MobileNetV3/sampling.py:858-910, :924-957); all
     * zero / NULL = the NAS-Bench-201 tree's total_grad_fn (sampling.py:502-505).
     *   cscale_steps  HOST [n_steps] or NULL: the weight of step i instead of classifier_scale
     *                 (weight_scheduling: scale * 0.1 ** t_i, computed by the host mirror like the reference, in double)
     *   guide_steps   HOST [n_steps] or NULL: 0 = this step uses the plain score (t outside [t_spot_end, t_spot]);
     *                 the surrogate is not evaluated at all on such a step
     *   weight_ratio  0 none | 1 per sample: w *= ||score_b||_p / ||grad_b||_p (prod_w: / ||w grad_b||_p)
     *                 | 2 per element: w *= |score| / |grad| (weight_ratio_abs)
     *   weight_ratio_p  the norm's p (1 in the reference's callers) */
    const float *cscale_steps;
    const int32_t *guide_steps;
    int32_t weight_ratio;
    float weight_ratio_p;
    int32_t prod_w;
    /* Chain recording of the MobileNetV3 sampler (reference MobileNetV3/sampling.py:980-988, :1030-1034), NULL: off.
     * After PC step i the first chain_keep graphs of (chain_denoise ? x_mean : x) -- raw, the caller applies its
     * inverse scaler -- are copied to slot chain_steps - 1 - (i * chain_steps) / chain_n of chain_x
     * [chain_steps][chain_keep][N][V] (device); later steps overwrite earlier ones of the same slot, as in the
     * reference, so only the last step of every slot is copied.  chain_n = sde.N (0: n_steps); slots no step maps to
     * (a truncated run) are left untouched: pass a zeroed buffer. */
    float *chain_x;
    int32_t chain_steps;
    int32_t chain_keep;
    int32_t chain_n;
    int32_t chain_denoise;
    /* n_steps_each of the reference's configs (sampling.py:244 `for i in range(n_steps)`, :283-293): corrector updates
     * per PC step, each with a score evaluation and a noise draw of its own.  0 means 1.  With c = corrector_steps the
     * corrector tape and the corrector records (rec_score_c, rec_total_c, rec_cgrad_c, rec_x_corr, rec_step_size) hold
     * n_steps * c entries, entry i * c + k for update k of PC step i; Philox draw indices are 1 + i (c + 1) + k for the
     * corrector updates and 1 + i (c + 1) + c for the predictor (c = 1: the numbering of earlier versions). */
    int32_t corrector_steps;
    /* Exact-global-batch Langevin step size for a batch sharded over several processes ("optional").
     * The step size of sampling.py:249-252 uses batch means of ||grad_b|| and ||noise_b||; by default every shard is its
     * own batch.  With langevin_exchange set, each corrector update (i) writes this shard's per-block sums to
     * langevin_local [dnag_langevin_partials_floats(cate_dims, batch)], (ii) calls
     *     langevin_exchange(langevin_user, stream)      -- host callback, returns 0 on success
     * which must enqueue, on `stream`, an all-gather of every rank's langevin_local into langevin_global
     * [langevin_world][that many floats] in rank order (RCCL all_gather; 8 KB per rank at 8192 graphs), and (iii) reduces
     * langevin_global in the fixed order a single batch of langevin_world * batch graphs would use: the step size is then
     * bit-identical to the unsharded run's.  All shards must hold the same number of graphs, a multiple of 8 (NAS-Bench-201)
     * or 3 (OFA).  NULL: per-shard batch means, no callback.  The callback runs on the calling thread while
     * dnag_pc_sample holds the device's second-stream lock (header comment, "Streams"): it must not call
     * dnag_pc_sample or dnag_cate_dsm_step for the same device. */
    dnag_langevin_exchange_fn langevin_exchange;
    void *langevin_user;
    float *langevin_local;             /* device */
    float *langevin_global;            /* device */
    int32_t langevin_world;
} dnag_pc_args;
size_t dnag_langevin_partials_floats(const dnag_cate_dims *cd, int batch);

int dnag_pc_sample(const dnag_pc_args *args);
/* sizeof(dnag_pc_args) / sizeof(dnag_train_args) as this library was built: a binding checks its own struct against
 * them before the first call (no counterpart in the reference: its boundary is Python objects). */
size_t dnag_pc_args_bytes(void);
size_t dnag_train_args_bytes(void);

/* ---- DSM training step (BASELINE config 3) --------------------------------------------------------
 * Replaces get_sde_loss_fn_nas(...).loss_fn (losses.py:127-166, likelihood_weighting=False) followed by
 * loss.backward() (losses.py:222-223): the denoising-score-matching loss of one batch and its gradient
 * wrt every CATE parameter, with the network in training mode (dropout active, models/utils.py:66-68).
 * The optimiser step, gradient clipping and EMA (losses.py:22-36, models/ema.py) are dnag_optim_step below.
 *
 * `params` / `grads` are flat fp32 device buffers in state_dict() order (dnag_cate_state_floats); `tables`
 * is the device copy of the constants dnag_cate_pack takes.  The random draws are inputs: z ~ N(0, 1)
 * (losses.py:141), and per graph the time draw folded through sde.marginal_prob (losses.py:139, :143) by
 * the host mirror: mean_coef (VPSDE exp(lmc), VESDE 1), std, and the time condition `labels` (999 t |
 * sigma(t), models/utils.py:93, :109).  `keep` holds one byte per dropout site element in the layout
 * below (NULL or dropout_p == 0: no dropout, which is also the eval-mode loss of losses.py:228-233):
 *     rows = batch * N, row = graph * N + node
 *     emb [rows][d_model]                                                   (cate.py:147)
 *     then per layer l:  [rows][n_head * N + d_model + d_ff + d_model] =
 *       attention probabilities of this query [n_head][N] (transformer.py:19) | self-attention block
 *       output (:80) | FFN hidden (:54) | FFN output (:108)
 * dnag_dropout_fill draws it on the device (Philox4x32-10).  grads == NULL: loss only. */
typedef struct dnag_train_args {
    const dnag_cate_dims *cate_dims;
    const float *params;               /* device [dnag_cate_state_floats] */
    const float *tables;               /* device [dnag_cate_table_floats] */
    int32_t batch;
    const float *x;                    /* device [batch][N][V] clean data (datasets_nas.py:11-18 scaling) */
    const float *z;                    /* device [batch][N][V] */
    const float *mean_coef;            /* device [batch] */
    const float *std;                  /* device [batch] */
    const float *labels;               /* device [batch] */
    const float *mask;                 /* device [batch][N][N] (mask_stride 1) or [1][N][N] (0) */
    int32_t mask_stride;
    int32_t score_mode;                /* DNAG_SCORE_VP or DNAG_SCORE_VE */
    int32_t reduce_mean;               /* losses.py:151-154 */
    const uint8_t *keep;               /* device [dnag_dropout_mask_bytes] or NULL */
    float dropout_p;
    float *loss;                       /* device [1] */
    float *grads;                      /* device [dnag_cate_state_floats] or NULL */
    void *workspace;
    size_t workspace_bytes;
    void *stream;
} dnag_train_args;

size_t dnag_train_workspace_bytes(const dnag_cate_dims *d, int batch);
size_t dnag_dropout_mask_bytes(const dnag_cate_dims *d, int batch);
int    dnag_dropout_fill(const dnag_cate_dims *d, int batch, float p, uint64_t seed, uint8_t *keep, void *stream);
int    dnag_cate_dsm_step(const dnag_train_args *args);

/* ---- optimiser side of the training step --------------------------------------------
 * Replaces, on flat fp32 buffers in state_dict() order, what follows loss.backward() in step_fn (losses.py:222-226):
 *   optimization_manager(...).optimize_fn (losses.py:22-36): warm-up learning rate, torch.nn.utils.clip_grad_norm_ over
 *   ALL parameters (total L2 norm, coefficient min(1, max_norm / (norm + 1e-6))), torch.optim.Adam.step (get_optimizer,
 *   losses.py:10-19: betas (beta1, 0.999), eps, weight_decay added to the gradient; no amsgrad), then
 *   ExponentialMovingAverage.update (models/ema.py:24-42): shadow -= (1 - decay) (shadow - param), with this operation
 *   order and no fused multiply-add, so that the shadow is bit-identical to the reference's given the same parameters.
 * Two launches: a fixed-order sum of squares (per-block partials in the workspace), then one element-wise pass in which
 * every block re-reduces the partials in the same order (no atomics: the result does not depend on the launch).
 * The scalars that the reference computes in Python doubles are computed by the host mirror the same way and handed
 * over as the fp32 values torch would round them to (diffusionnag_amd/losses.py FusedAdam):
 *   step_size = lr_t / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t), lr_t = lr min(step / warmup, 1), t = Adam step count.
 * flags: DNAG_OPTIM_ADAM and / or DNAG_OPTIM_EMA.  grads are read only (the reference scales .grad in place when it
 * clips; nothing reads it afterwards).  grad_norm_out (may be NULL) receives the total norm before clipping. */
#define DNAG_OPTIM_ADAM 1
#define DNAG_OPTIM_EMA  2
typedef struct dnag_optim_args {
    size_t n;                          /* parameters (dnag_cate_state_floats for the score network) */
    int32_t flags;
    float *params;                     /* device [n], updated in place */
    const float *grads;                /* device [n] */
    float *exp_avg, *exp_avg_sq;       /* device [n] Adam moments, updated in place */
    float *shadow;                     /* device [n] EMA shadow parameters (DNAG_OPTIM_EMA) */
    float beta1, beta2;                /* as torch rounds the python floats to fp32 */
    float one_minus_beta1, one_minus_beta2;
    float eps, weight_decay;
    float step_size, bc2_sqrt;
    float grad_clip;                   /* max_norm; negative: no clipping (losses.py:32) */
    float ema_one_minus_decay;         /* 1 - min(decay, (1 + n) / (10 + n)) in double, rounded to fp32 (ema.py:34-38) */
    float *grad_norm_out;              /* device [1] or NULL */
    void *workspace;                   /* dnag_optim_workspace_bytes() */
    size_t workspace_bytes;
    void *stream;
} dnag_optim_args;
size_t dnag_optim_workspace_bytes(void);
size_t dnag_optim_args_bytes(void);
int    dnag_optim_step(const dnag_optim_args *args);

/* ---- exit of a sampling round, NAS-Bench-201 ---------------------------------
 * Replaces quantize (utils.py:244-262) + is_valid_NAS201_x (analysis/arch_functions.py:284-329) + the argmax
 * decode (:209-216) that BasicArchMetricsMeta.compute_validity (:95-118) runs per sample on the host, and the
 * `set(all_samples)` bookkeeping of generate_archs_meta (main_exp/diffusion/run_lib.py:141-152).
 *   x [batch][8][7]  sampler output ((x_mean + 1) / 2 scale); thresholded at 0.5 here
 *   status [batch]   -1 valid | 1 MULTIPLE_NODE_TYPES | 2 No_START_END | 3 INTERM_START_END (:308-313)
 *   code [batch]     sum_i (argmax(row i+1) - 2) 5^i, i = 0..5 (six cell edges, operation types 2..6), -1 if invalid
 *   first_seen [DNAG_NB201_NCODES] (may be NULL) persistent across rounds: the smallest index_base + graph index
 *                    that produced each code; dnag_nb201_seen_reset sets every entry to DNAG_NB201_UNSEEN
 *   counts [2] (may be NULL): valid graphs of this call, distinct codes seen so far.
 * All pointers are device pointers; nothing synchronises. */
#define DNAG_NB201_NCODES 15625
#define DNAG_NB201_UNSEEN 0x7fffffff
int dnag_nb201_seen_reset(int32_t *first_seen, void *stream);
int dnag_nb201_validity(const float *x, int batch, int32_t *status, int32_t *code, int32_t *first_seen,
                        int32_t index_base, int32_t *counts, void *stream);

/* Exit of a MobileNetV3 / OFA sampling round: threshold (reference MobileNetV3/utils.py:254-255) and the validity
 * rules of is_valid_OFA_x (MobileNetV3/analysis/arch_functions.py:374-429) for x [batch][20][9] on the device.
 *   status [batch]      -1 valid | 1 MULIPLE_NODES | 2 INVALID_1OR2_LAYERS | 3 INVALID_3AND4_LAYERS
 *   types  [batch][20]  type 0..8 of every layer slot (first 1 of its row), -1 for a skipped layer: the input of
 *                       get_string_from_onehot_x (:135-144); written for every graph
 *   counts [4] (may be NULL)  graphs with status -1, 1, 2, 3 */
int dnag_ofa_validity(const float *x, int batch, int32_t *status, int8_t *types, int32_t *counts, void *stream);

/* ---- exit: threshold (utils.py:244-262) --------------------------------------------------------
 * q[i] = x[i] >= 0.5 ? 1 : 0, n elements, device pointers. */
int dnag_quantize(const float *x, float *q, size_t n, void *stream);

/* ---- kernel timing for bench.py ----------------------------------------------------------------
 * While enabled, every `every`-th launch of the head GEMM (the dominant kernel) is bracketed by a
 * pair of HIP events on the launch stream.  dnag_timing_read() synchronises those events and
 * returns the number of timed launches, their mean duration (ms) and the FLOPs of one launch. */
int dnag_timing_begin(int every, int max_samples);
int dnag_timing_read(int *n_samples, double *mean_ms, double *flops_per_launch);
int dnag_timing_end(void);

#ifdef __cplusplus
}
#endif
#endif /* DNAG_H */
