/*
 * trace_hip.h -- C ABI of libtrace_hip.so: the MI355X (gfx950) kernels of the TRACE guided
 * trajectory-diffusion sampler.
 *
 * The reference (nv-tlabs/trace) has no FFI: its hot path is Python calling ATen.  Each entry
 * point below replaces the ATen op group of one reference function (cited as
 * tbsim/...:line, relative to the reference root) and is what a binding for this path would
 * bind (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless named *_host
 *   - all tensors fp32, row-major, contiguous unless a stride argument says otherwise
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue
 *   - no hidden device allocation after trace_unet_create(): scratch comes in as `workspace`
 *   - return 0 on success, negative TRACE_E* otherwise; trace_last_error() holds the text
 *   - M = trajectories (agents x samples), trajectory index m = agent * num_samp + sample
 *     (tbsim/utils/tensor_utils.py:664-677), T = horizon, state layout (x,y,v,yaw,acc,yawvel)
 */
#ifndef TRACE_HIP_H
#define TRACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRACE_OK 0
#define TRACE_EINVAL (-1)   /* bad argument / unsupported shape */
#define TRACE_EHIP (-2)     /* a HIP runtime call failed */
#define TRACE_ENOMEM (-3)   /* workspace too small */

const char* trace_last_error(void);
int trace_abi_version(void);

/* ------------------------------------------------------------------ denoiser ---- */
/* Temporal UNet hyper-parameters (tbsim/models/temporal.py:49-115). */
typedef struct {
  int horizon;        /* 52 */
  int in_channels;    /* 6 + map_grid_feature_dim = 38 */
  int n_levels;       /* 3 */
  int dims[4];        /* channels per level: 64,128,256 (first n_levels entries used) */
  int cond_dim;       /* 256 */
  int time_dim;       /* 32 */
  int n_groups;       /* 8 */
  int n_steps;        /* diffusion steps: 100 */
} trace_unet_config;

typedef struct trace_unet trace_unet;

/* Number of fp32 parameters trace_unet_create expects, for `cfg`. */
int64_t trace_unet_param_count(const trace_unet_config* cfg);

/* Builds the denoiser from a flat HOST array of parameters in canonical order
 * (documented in trace_amd/unet_params.py: residual blocks in execution order, then
 * down/up-sample convs, final block, 1x1 head, time MLP).  Packs the weights into
 * MFMA-fragment order for the forward and the input-VJP and uploads them.
 * Replaces TemporalMapUnet.__init__ + load_state_dict (tbsim/models/temporal.py:51-115). */
int trace_unet_create(const trace_unet_config* cfg, const float* params_host, int64_t n_params,
                      trace_unet** out);
void trace_unet_destroy(trace_unet* u);

/* Floats per conditioning row in the table written by trace_unet_cond_table
 * (sum of C_out over the residual blocks). */
int trace_unet_cond_width(const trace_unet* u);

/* table[r, :] = W_c * mish(cond[r, :]) + b for the per-block Linear layers' cond part
 * (the `cond` columns of time_mlp.1 in ResidualTemporalMapBlockConcat,
 * tbsim/models/temporal.py:22-26,44; the time columns are pre-tabulated per step at create
 * time since t is shared by all rows).  cond: (n_rows, cond_dim). */
int trace_unet_cond_table(const trace_unet* u, const float* cond, int n_rows, float* table,
                          void* stream);

/* Workspace bytes for `rows` denoiser rows; `save` != 0 keeps what the input-VJP needs. */
int64_t trace_unet_workspace_bytes(const trace_unet* u, int rows, int save);

/* One denoiser evaluation, TemporalMapUnet.forward (tbsim/models/temporal.py:117-142).
 *   x_in   (rows, in_channels padded to 64, horizon)  channel-major, as written by
 *          trace_map_query; pad channels must be zero
 *   cond_table (n_cond_rows, cond_width); row r uses table row r / rows_per_cond
 *   step   diffusion step t (same for all rows, tbsim/models/trace.py:738)
 *   pred   (rows, horizon, 2)
 * With classifier-free guidance the caller stacks conditional and unconditional rows
 * (rows = 2M) so both passes of tbsim/models/trace.py:572,588 run as one call.
 * Calls of 4096 rows and more launch half of their rows on a second stream owned by `u` (joined back
 * into `stream` before the call returns control of it; not under stream capture): one call at a
 * time per trace_unet.
 * A row's prediction (and its input gradient below) does not depend on how many rows the call has: groups of 16
 * consecutive rows share operand scales, so any call whose rows start at a multiple of 16 reproduces its rows of a
 * larger call bit for bit, whichever tile shape the launch sizes select (tests/test_gpu_full_size.py). */
int trace_unet_forward(trace_unet* u, const float* x_in, const float* cond_table,
                       int rows_per_cond, int step, int rows, float* pred, void* workspace,
                       int64_t workspace_bytes, int save, void* stream);

/* Input-VJP of the last trace_unet_forward(save=1) on the same workspace: the autograd
 * backward the reference runs for clean guidance (tbsim/models/trace.py:630,640-649),
 * restricted to the trajectory channels the sampler uses (:674-676).
 *   grad_pred (rows, horizon, 2)
 *   grad_x    (rows, horizon, 2): gradient w.r.t. input channels [4,5] (acc, yawvel). */
int trace_unet_backward_input(trace_unet* u, const float* grad_pred, int rows, float* grad_x,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* Per-launch timing of the implicit-GEMM conv kernels (the dominant kernel): between begin and
 * end every conv launch is bracketed by HIP events on its stream.  end() synchronises on the
 * events and returns the summed kernel time, the number of launches and their algorithmic FLOPs
 * (2 * rows * L_out * C_out * C_in * taps, structural zeros of transposed convs not counted).
 * No counterpart in the reference (it has no profiling hooks); used by
 * bench.py for the roofline figures. */
int trace_profile_begin(int max_launches);
int trace_profile_end(double* total_ms, int64_t* launches, double* total_flops);
/* Matrix-pipe time of the profile trace_profile_end last closed, had every conv launch run at the chip's dense
 * MFMA peak: sum over launches of (algorithmic FLOPs x MFMA terms per product: 3 for the fp16-pair scheme, 6 for
 * the bf16-triple scheme) / 2500 TFLOP/s, or FLOPs / 157.3 TFLOP/s for launches on the fp32 MFMA.  The roofline
 * fraction of bench.py is this over the measured time.  No counterpart in the reference. */
double trace_profile_pipe_seconds(void);

/* Kernel choice for the full-chip multi-tap convolutions of the denoiser: 0 = conv_bf3_kernel (two workgroups
 * per CU, the default), 1 = conv_ws_kernel (persistent, wave-specialised: csrc/conv_ws_kernels.h), -1 = leave
 * as is.  Both give the same bits; the switch exists for A/B measurements and tests.  Initial value 0 (the library
 * reads no environment variable).  Returns the previous value.  No counterpart in the reference. */
int trace_conv_ws_mode(int mode);

/* Operand scheme of the denoiser's implicit-GEMM convolutions (fp32 results either way): 1 (default) = every fp32
 * operand as TWO fp16 terms, three v_mfma_f32_16x16x32_f16 per product, the operands brought into fp16's exponent
 * range by exact powers of two (weights: per layer on the host; activations: per 16 trajectories from the largest
 * |value| the producing kernel recorded) -- used by every launch whose inputs all carry such a scale, the others
 * stay on 0; 0 = three bf16 terms, six v_mfma_f32_16x16x32_bf16 per product everywhere; -1 = leave as is.  The
 * wave-specialised kernel (trace_conv_ws_mode) only runs under scheme 0.  Initial value 1 (the library reads no environment
 * variable).  Returns the previous value.  No counterpart in the
 * reference (tbsim/models/temporal.py runs torch fp32 convolutions). */
int trace_conv_scheme(int mode);

#ifdef TRACE_CONV_STAMPS
/* Diagnostic builds only (scripts/stamps.py builds a second library with -DTRACE_CONV_STAMPS): conv launch
 * number `launch_index` after this call writes 16 s_memtime stamps per workgroup into `buf` (device). */
void trace_debug_stamps(void* buf, int launch_index);
#endif

/* ------------------------------------------------------------- sampler steps ---- */
/* Constants of the sampler: normalisation (tbsim/models/trace.py:415-447), unicycle bounds
 * (tbsim/dynamics/unicycle.py:7-18), dt. */
typedef struct {
  float norm_add[6];
  float norm_div[6];
  float dt;
  float acc_lo, acc_hi;   /* acce_bound */
  float v_lo, v_hi;       /* vbound */
  float max_steer;
  float max_yawvel;
} trace_dyn_params;

/* Per-step scalars looked up from the schedule tables (tbsim/models/trace.py:170-203). */
typedef struct {
  float sqrt_recip_alphas_cumprod;
  float sqrt_recipm1_alphas_cumprod;
  float posterior_mean_coef1;
  float posterior_mean_coef2;
  float posterior_log_variance_clipped;
  float sqrt_alphas_over_one_minus_alphas_cumprod;
  float sqrt_recip_one_minus_alphas_cumprod;
  int t;                  /* diffusion step; t == 0 disables noise and guidance */
} trace_step_coefs;

/* query_map_feats + concat + transpose (tbsim/models/trace.py:341-368,570;
 * tbsim/models/trace_helpers.py:270-319; tbsim/utils/geometry_utils.py:105-171).
 *   x (M, T, 6) scaled;  grid (n_agents, Hg, Wg, C) CHANNELS-LAST (the reference permutes to
 *   this layout at trace_helpers.py:293 on every call; here the caller does it once per sampling
 *   call), trajectory m reads grid m / num_samp (no per-sample copy, cf. trace.py:724);
 *   raster_from_agent (n_agents, 3, 3);  x_in (M, c_pad, T): channels 0..5 = x, 6..6+C-1 =
 *   bilinear map features; channels beyond are left untouched (must be zero). */
int trace_map_query(const float* x, const float* grid, const float* raster_from_agent,
                    float* x_in, int M, int num_samp, int T, int C, int Hg, int Wg, int c_pad,
                    float img_w, float img_h, const trace_dyn_params* p, void* stream);
/* The same for both halves of a classifier-free call in ONE launch: the reference queries the conditioned and
 * the unconditioned grid one after the other (tbsim/models/trace.py:570,578-583); rows 0..M-1 of x_in take
 * `grid`, rows M..2M-1 `grid_non_cond` (same x, same bilinear weights). */
int trace_map_query_cfg(const float* x, const float* grid, const float* grid_non_cond,
                        const float* raster_from_agent, float* x_in, int M, int num_samp, int T, int C,
                        int Hg, int Wg, int c_pad, float img_w, float img_h, const trace_dyn_params* p,
                        void* stream);

/* Unicycle action->state rollout, mode='parallel'
 * (tbsim/models/trace_helpers.py:387-475; wrapper tbsim/models/trace.py:390-413).
 *   actions (M, T, 2), scaled_input != 0: descale channels [4,5] first
 *   curr_states (n_agents, 4), trajectory m uses row m / num_samp
 *   out (M, T, 6) = [states, descaled actions]; scaled_output != 0: scale all 6 channels. */
int trace_dyn_rollout(const float* actions, const float* curr_states, float* out, int M,
                      int num_samp, int T, int scaled_input, int scaled_output,
                      const trace_dyn_params* p, void* stream);

/* Adjoint of trace_dyn_rollout(scaled_input=1, scaled_output=0): given the gradient w.r.t. the
 * descaled (M,T,6) trajectory returns the gradient w.r.t. the scaled actions (M,T,2)
 * (autograd through tbsim/models/trace_helpers.py:431-475 and trace.py:399). */
int trace_dyn_adjoint(const float* actions, const float* curr_states, const float* grad_traj,
                      float* grad_actions, int M, int num_samp, int T, const trace_dyn_params* p,
                      void* stream);

/* Classifier-free-guidance combine, op-for-op as tbsim/models/trace.py:591-598 (no FMA
 * contraction).  pred_c, pred_u (M,T,2); x (M,T,6); x0 (M,T,2).
 * w == 0: x0 = pred_c (pred_u may be NULL). */
int trace_cfg_combine(const float* pred_c, const float* pred_u, const float* x, float* x0, int M,
                      int T, float w, const trace_step_coefs* c, void* stream);

/* Scales the guidance gradient w.r.t. x0 into gradients w.r.t. the two denoiser outputs
 * (chain rule through tbsim/models/trace.py:591-596 with x_t detached). */
int trace_cfg_combine_adjoint(const float* grad_x0, float* grad_pred_c, float* grad_pred_u, int M,
                              int T, float w, const trace_step_coefs* c, void* stream);

/* q_posterior mean (tbsim/models/trace.py:552-559): mean = c1*x0 + c2*x[...,4:6]. */
int trace_posterior_mean(const float* x0, const float* x, float* mean, int M, int T,
                         const trace_step_coefs* c, void* stream);

/* End of p_sample (tbsim/models/trace.py:653-704): applies guidance + noise and rolls the
 * dynamics out again.  clean != 0: x0' = x0 - nz*grad; mean = c1*x0' + c2*x_t; a = mean + nz*sigma*noise
 * (:683-693); else a = (c1*x0 + c2*x_t) - nz*grad + nz*sigma*noise (:695).  grad may be NULL.
 * a_out (M,T,2) (optional) receives the scaled actions before the rollout; x_next (M,T,6). */
int trace_posterior_step(const float* x0, const float* grad, const float* x, const float* noise,
                         const float* curr_states, float* x_next, float* a_out, int M,
                         int num_samp, int T, int clean, const trace_step_coefs* c,
                         const trace_dyn_params* p, void* stream);
/* The same with the input gradient of a classifier-free call given as its two halves (the reference's autograd
 * sums the gradients of the conditioned and the unconditioned denoiser pass into x, tbsim/models/trace.py:578-596,
 * 674-676): grad = grad_cond + grad_non_cond, formed here instead of by a kernel of its own. */
int trace_posterior_step_cfg(const float* x0, const float* grad_cond, const float* grad_non_cond,
                             const float* x, const float* noise, const float* curr_states, float* x_next,
                             float* a_out, int M, int num_samp, int T, int clean, const trace_step_coefs* c,
                             const trace_dyn_params* p, void* stream);

/* ------------------------------------------------------ guidance objectives ---- */
/* All objectives read the descaled trajectory traj (M,T,6), ADD their weighted gradient into
 * grad_traj (M,T,6) (may be NULL: value only) and write the unweighted per-trajectory loss
 * into loss (M) for trajectories of active agents (others untouched).  Per-agent tables have
 * n_agents rows; `scale[a]` = weight / (|mask| * num_samp) is the factor of
 * DiffuserGuidance.compute_guidance_loss (tbsim/utils/guidance_loss.py:919); NEGATIVE = agent not covered
 * (loss untouched, no gradient); 0 = covered with weight 0 (loss written, zero gradient). */

/* TargetPosLoss (guidance_loss.py:523-540): target (n_agents,2), t_min (n_agents) first step. */
int trace_guide_target_pos(const float* traj, float* grad_traj, float* loss, const float* target,
                           const int* t_min, const float* scale, int M, int num_samp, int T,
                           void* stream);

/* TargetPosAtTimeLoss (guidance_loss.py:488-501): target (n_agents,2), t_at (n_agents). */
int trace_guide_target_pos_at_time(const float* traj, float* grad_traj, float* loss,
                                   const float* target, const int* t_at, const float* scale,
                                   int M, int num_samp, int T, void* stream);

/* compute_progress_loss (guidance_loss.py:542-590) used by the Global* objectives.
 * goal (n_agents): goal distance; mode[a] = 0: relu(final_dist - goal) (:573),
 * 1: relu(goal - (init_dist - final_dist)) (:585-588). */
int trace_guide_progress(const float* traj, float* grad_traj, float* loss, const float* target,
                         const float* goal, const int* mode, const float* scale, int M,
                         int num_samp, int T, void* stream);

/* TargetSpeedLoss / MinSpeedLoss (guidance_loss.py:148-196).  kind 0: mean_t (v - s)^2 on the
 * speed channel; 1: same on finite-difference speed; 2: mean_t gelu(s - fd speed)^2. */
int trace_guide_speed(const float* traj, float* grad_traj, float* loss, const float* speed,
                      const float* scale, int kind, float dt, int M, int num_samp, int T,
                      void* stream);

/* AgentCollisionLoss (guidance_loss.py:264-338) + transform_agents_to_world
 * (tbsim/models/trace_helpers.py:343-364): all pairs inside each scene.
 *   world_from_agent (n_agents,3,3); radius (n_agents); disk_x (n_agents, num_disks);
 *   scene_start (n_agents): first agent of the agent's scene, scene_size (n_agents);
 *   in_mask (n_agents): 1 if the agent is guided (rows kept, gradient allowed);
 *   workspace: M*T*4 floats (world poses). */
int trace_guide_agent_collision(const float* traj, float* grad_traj, float* loss,
                                const float* world_from_agent, const float* radius,
                                const float* disk_x, int num_disks, float buffer_dist,
                                const int* scene_start, const int* scene_size,
                                const int* in_mask, const float* scale, int n_agents,
                                int num_samp, int T, float* workspace, void* stream);

/* trace_guide_agent_collision for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same arguments and results as trace_guide_agent_collision minus `workspace`: the world poses go to a
 * static device array of trace_static_scratch_bytes() (8 MiB) that every GPU the library runs on holds
 * until the process ends.  Calls no hipMalloc / hipFree or allocating routine.  Rejects
 * n_agents * num_samp * T * 4 floats beyond that size with TRACE_EINVAL.  All calls on one GPU share
 * the array: calls that overlap on one GPU (two streams) race; calls in stream order do not.
 * trace_static_scratch_init() resolves the array's device address for the CURRENT device (once per
 * device; the entry point does it on first use otherwise) -- call it before a graph capture. */
int trace_static_scratch_init(void);
int64_t trace_static_scratch_bytes(void);
int trace_guide_agent_collision_static(const float* traj, float* grad_traj, float* loss,
                                       const float* world_from_agent, const float* radius,
                                       const float* disk_x, int num_disks, float buffer_dist,
                                       const int* scene_start, const int* scene_size,
                                       const int* in_mask, const float* scale, int n_agents,
                                       int num_samp, int T, void* stream);

/* MapCollisionLoss (guidance_loss.py:382-464; tbsim/utils/metrics.py:216-240).
 *   drivable (n_agents, H, W) uint8 (bool); extent (n_agents,3); raster_from_agent (n_agents,3,3)
 *   local_pts (n_pts, 2): unit-box sample grid, cartesian_prod(linspace(-.5,.5,n_l),
 *   linspace(-.5,.5,n_w)) as built at guidance_loss.py:356-358; n_pts <= 128
 *   workspace: M*T floats (per-step losses). */
int trace_guide_map_collision(const float* traj, float* grad_traj, float* loss,
                              const uint8_t* drivable, const float* extent,
                              const float* raster_from_agent, const float* local_pts, int n_pts,
                              const float* scale, int H, int W, int M, int num_samp, int T,
                              float* workspace, void* stream);

/* The same call with a diagnostic output for the parity tests (no counterpart in the reference, whose
 * `torch.amin` over the cdist matrix at guidance_loss.py:444-458 keeps its arg-min to itself): argmin
 * (M*T, n_pts, 2) int32 receives, for every sample point that is off-road in an active step, the on-road sample
 * point(s) the loss paired it with -- {first arg-min row | number of tied rows << 8, the 2nd .. 4th tied row one per
 * byte (0xff = none)}; `torch.amin` shares the gradient between exactly tied rows, which the regular 10 x 10 sample grid
 * produces -- and {-1, -1} elsewhere.  The
 * reference's matmul-form cdist makes that choice chaotic at the ulp level; given the
 * choice, loss and gradient are smooth, which is what tests/test_gpu_parity.py checks on every trajectory-step. */
int trace_guide_map_collision_argmin(const float* traj, float* grad_traj, float* loss,
                                     const uint8_t* drivable, const float* extent,
                                     const float* raster_from_agent, const float* local_pts, int n_pts,
                                     const float* scale, int H, int W, int M, int num_samp, int T,
                                     float* workspace, int* argmin, void* stream);

/* SocialGroupLoss (guidance_loss.py:744-814): every member of a group keeps `social_dist` to one
 * assigned neighbour of its group per (sample, step) -- the nearest member, or the member the caller
 * drew at random (the reference draws torch.randint / torch.rand inside the loss, :803-804; here the
 * draw is an argument so that it can be injected).  The leader's positions are constants (:775-776).
 *   members (n_agents, group_width) int32: global agent indices of the agent's group in group order, -1 padded
 *     (rows of agents outside any group: all -1, scale 0); group_width >= the largest group
 *   is_leader (n_agents) int32; pick (M, T) int32 or NULL: >= 0 = global index of the assigned
 *     neighbour, -1 = nearest
 *   workspace: M*T*4 floats (world poses); nbr_ws: M*T int32; dist_ws: M*T floats
 * loss[m] = mean_t (dist - social_dist)^2.  Groups of up to TRACE_SOCIAL_GROUP_WIDTH members take the exact
 * distance, larger ones the matmul form torch.cdist switches to above 25 points (guidance_loss.py:786,
 * SURVEY.md appendix A), evaluated in ATen's operation order. */
#define TRACE_SOCIAL_GROUP_WIDTH 25
int trace_guide_social_group(const float* traj, float* grad_traj, float* loss,
                             const float* world_from_agent, const int* members, int group_width,
                             const int* is_leader, const int* pick, float social_dist,
                             const float* scale, int n_agents, int num_samp, int T,
                             float* workspace, int* nbr_ws, float* dist_ws, void* stream);

/* Agent-centric drivable raster of the closed-loop environment (the map
 * crop + rotate that trajdata's collation does on the host for the reference,
 * trajdata/src/trajdata/data_structures/collation.py:17-255, and the drivable layer of
 * tbsim/utils/trajdata_utils.py:85-132), for synthetic scenes whose obstacles are axis-aligned boxes.
 *   pose (n_agents, 3) world x, y, yaw;  scene_of_agent (n_agents) int32 row into boxes;
 *   boxes (n_scenes, n_boxes, 4) centre x, y, half sizes;  raster_from_agent_host: 9 floats in HOST
 *   memory (shared by all agents, trajdata_utils.py:196-200; must be scale + offset)
 *   image (n_agents, channels, H, W) float or NULL: channel 0 = drivable, 1 = obstacle, others 0;
 *   drivable (n_agents, H, W) uint8 or NULL.  W % 4 == 0, n_agents <= 65535. */
int trace_env_rasterize(const float* pose, const int* scene_of_agent, const float* boxes, int n_boxes,
                        const float* raster_from_agent_host, float* image, int channels,
                        uint8_t* drivable, int n_agents, int H, int W, void* stream);

/* ---- map encoder: the layers of the reference's MapEncoder -- torchvision ResNet-18
 * trunk behind RasterizedMapEncoder (tbsim/models/base_models.py:402-491) and MapGridDecoder
 * (tbsim/models/trace_helpers.py:231-268; Up / DoubleConv / ConvBlock at base_models.py:169-260) -- as kernels.
 * Activations are NHWC fp32 (B, H, W, ctot); a layer reads / writes the channel window [coff, coff + C) of its
 * buffer, so the decoder's skip concatenations (base_models.py:224) are written in place.
 *
 * trace_conv2d: out = act(scale[co] * conv(in, w) + shift[co] (+ res)), cross-correlation as torch.nn.Conv2d.
 *   evaluation-mode BatchNorm and the conv bias are folded into scale / shift by the caller (NULL = 1 / 0).
 *   packed_w: the weight (cout, cin, kh, kw) split exactly into three bf16 planes and laid out in MFMA fragment
 *   order [cout/16][cin/32][kh*kw][3 planes][64 lanes][8] (trace_amd/map_encoder.py pack_conv2d_weight);
 *   fp32 products through 6 bf16 MFMA terms as in the denoiser.  cin % 32 == 0, cout % 16 == 0.
 *   Built shapes: 3x3 s1 p1 at output widths 56/28/14/7, 3x3 s2 p1 and 1x1 s2 at 28/14/7, 1x1 s1 at 56 (a 224 x
 *   224 raster); anything else returns TRACE_EINVAL. */
int trace_conv2d(const float* in, int B, int H, int W, int in_ctot, int in_coff, int cin,
                 const void* packed_w, int kh, int kw, int stride, int pad, float* out, int out_ctot,
                 int out_coff, int cout, const float* scale, const float* shift, const float* res,
                 int res_ctot, int res_coff, int relu, void* stream);

/* ResNet stem: 7x7 stride-2 pad-3 conv of the NCHW raster (B, C <= 8, H, W) + BatchNorm + ReLU
 * (torchvision resnet conv1/bn1/relu; C-channel stem at base_models.py:437-441) -> NHWC (B, H/2, W/2, 64).
 *   w_tap_ci_co: the weight (64, C, 7, 7) permuted to (7, 7, C, 64), fp32. */
int trace_map_stem(const float* image, const float* w_tap_ci_co, const float* scale, const float* shift,
                   float* out, int B, int C, int H, int W, void* stream);

/* MaxPool2d(3, stride 2, padding 1) of an NHWC tensor (torchvision resnet maxpool). C % 4 == 0. */
int trace_maxpool3x3s2(const float* in, float* out, int B, int H, int W, int C, void* stream);

/* nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) (base_models.py:211) of NHWC (B, H, W, C)
 * into the channel window [out_coff, out_coff + C) of (B, 2H, 2W, out_ctot). */
int trace_upsample2x(const float* in, float* out, int B, int H, int W, int C, int out_ctot, int out_coff,
                     void* stream);

/* nn.LayerNorm over ALL n elements of each image with an elementwise affine (trace_helpers.py:258,267);
 * gamma / beta in the memory order of `in` (the caller permutes the (C, H, W) parameters to (H, W, C)). */
int trace_layernorm_image(const float* in, const float* gamma, const float* beta, float* out, int B, int n,
                          float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRACE_HIP_H */
