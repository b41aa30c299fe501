/* rgbm.h — C ABI of librgbm_hip.so (MI355X / gfx950).
 *
 * The reference (hyperplane-lab/RGBManip) is pure Python and has no FFI: its "plugin ABI" for this path is the
 * set of Python classes selected by Hydra name strings.  Each entry point below states which reference
 * interface it stands in for (paths relative to /root/reference).  The Python classes in rgbmanip_amd/ mirror
 * those interfaces and bind these functions through ctypes.
 *
 * Conventions: every function returns 0 on success, a negative code on error (rgbm_last_error() has the text);
 * all `*_dev` / device pointers are HIP device memory owned by the caller (e.g. the PyTorch caching allocator);
 * all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) with no hidden
 * synchronisation; handles own only their packed weights; a handle is not thread-safe, distinct handles are.
 * Forwards on different streams may overlap on the device (round 4: they differed intermittently from the one-stream result until the library was
 * built without packed fp32 instructions; tools/check_two_stream_forwards.py is the regression check, DESIGN.md section 5d has the finding).
 * dtype: 0 = fp32 (exact-fp32 MFMA, the parity gate), 1 = bf16 storage + fp32 accumulate (throughput mode).
 */
#ifndef RGBM_H_
#define RGBM_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGBM_VERSION 100
#define RGBM_F32 0
#define RGBM_BF16 1
#define RGBM_F16 2       /* IEEE half storage (saturating stores), fp32 accumulation; same kernels as RGBM_BF16 */
#define RGBM_BF16X3 3    /* split pairs: every value kept as bf16 hi + bf16 lo (16 significand bits, 4 bytes, fp32 tensor layout with
                            each 16-byte chunk of 4 channels holding {hi01, hi23, lo01, lo23}); products hi*hi + lo*hi + hi*lo on
                            the bf16 matrix pipe, fp32 accumulation: fp32-level results (1e-4 gate) at 3 MFMAs per product */

int rgbm_version(void);
const char* rgbm_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * AdaPose estimator network.
 * Replaces: StereoPoseNet_with_depth.__init__/load_state_dict and .forward
 *   models/pose_estimator/AdaPose/lib/network_v5.py:301-376, 418-519   (called from interface_v5.py:43-56, 279-280)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct rgbm_weight_desc {
  const char* name;        /* state_dict key, with or without the DataParallel "module." prefix */
  const float* data;       /* host fp32, contiguous, PyTorch layout */
  int ndim;
  const int64_t* shape;
} rgbm_weight_desc;

typedef struct rgbm_adapose rgbm_adapose_t;

typedef struct rgbm_adapose_out {   /* device fp32, shapes of the reference's output dict (network_v5.py:510-515) */
  float *view1_nocs, *view2_nocs;   /* [B,1024,3] */
  float *view1_depth, *view2_depth; /* [B,1024]   */
  float *view1_r, *view2_r;         /* [B,3,3]    */
  float *view1_t, *view2_t;         /* [B,3]      */
  float *view1_s, *view2_s;         /* [B,3]      */
} rgbm_adapose_out;

/* dtype: RGBM_F32 (parity mode), RGBM_BF16 (throughput mode) or RGBM_F16.
 * norm_mode: 0 = eval-mode BatchNorm3d folded into the convs (the default and the benchmarked path);
 *            1 = per-sample statistics: every BatchNorm3d of the cost-regularisation net normalises a view's volume with that
 *                volume's own biased mean / variance — what the reference as shipped computes (interface_v5.py:39-56 never calls
 *                .eval() and runs one pose per call), with Dropout2d as identity.  Generic kernels, materialised volume. */
int rgbm_adapose_create(rgbm_adapose_t** h, int device, const rgbm_weight_desc* w, int n_w, int dtype, int norm_mode);
int rgbm_adapose_destroy(rgbm_adapose_t* h);
/* views per cost-volume chunk (default 512 = batch 256 in one chunk); bounds the workspace */
int rgbm_adapose_set_chunk(rgbm_adapose_t* h, int max_chunk_views);
/* options: "max_chunk" (views per cost-volume chunk), "fuse_final" (bf16 only; 1 [default] = PSPNet's final 1x1 runs inside
 * up_3's kernel and the 64-channel up_3 output is never written; 0 = two launches), "sparse_tail" (bf16 + cost_impl 3 only; 1 [default] = conv11, the
 * skip add, the prob conv, softmax and depth are evaluated only on the 3x3 neighbourhoods of the chosen pixels and u11
 * is never written; 0 = dense conv11 + gathering prob kernel), "cost_impl" (3 = 2 with the depth-sweeping conv0 kernel [default for bf16;
 * fp32 nets run 2]; 2 = halo-tiled 3-D convs with the plane-sweep volume fused into conv0's loader; 1 = halo-tiled convs on a materialised volume; 0 = generic implicit
 * GEMM on a materialised volume), "igemm_conv6" (1 [default] = conv6 of the cost regularisation on the implicit-GEMM kernel), "upconv" (bit 0 / 1:
 * PSPUpsample up_1 / up_2 as a 1x1 GEMM at the low resolution + tap combination, bit 2: up_3 + `final` in one kernel from the
 * half-resolution tensor [16-bit and split-pair nets]; default 7; 0 = x2 resize followed by the 3x3 conv, the reference's operator
 * order), "stem" (1 [default for 16-bit and split-pair nets] = conv1 7x7 + ReLU + max-pool in one kernel from the NCHW images; 0 = padded
 * NHWC copy, implicit-GEMM conv, pool kernel — the only path that materialises the `conv1` tap of rgbm_adapose_fetch).
 * "sparse_dec" (sparse cost regularisation with the sparse tail: the probability volume is read only at the chosen pixels, so every 3-D
 * layer has a dependency cone per axis; 2 [default] = the plane sweep, conv1..conv5, conv7 and conv9 run only on the tiles inside those
 * cones — c0..c5 / u7 / u9 are undefined elsewhere, every network output is bit-identical; 1 = conv7 / conv9 only; 0 = dense).
 * "view2_heads" (1 [default] = all ten outputs; 0 = the probability volume, the point heads and the pose regression run for the
 * view-1 crops only and the five view-2 outputs are filled with NaN: what AdaPoseEstimator_v5.estimate consumes,
 * interface_v5.py:318-374, at about three quarters of the time; the backbone still runs on both views).
 * "sweep_f16" (bf16 nets with cost_impl 3 and upconv bit 2; 1 [default since round 5] = `final` writes the 32-channel feature map as f16
 * instead of bf16 - same bytes, three more mantissa bits - and the plane sweep (persistent conv0_sweep kernel) blends it with packed f16
 * FMAs and multiplies with f16 MFMAs; c0 and everything behind it stay bf16.  Features beyond +-65504 saturate, as in an fp16 net; 0 =
 * bf16 feature map and the fp32 blend of rounds 1-4.  The "feat" tap of rgbm_adapose_fetch converts from whichever form was written.
 * With 1 the one-kernel up_3 + `final` also runs its tap combination, PReLU and `final` in packed f16: faster, and closer to fp32).
 * Set before querying the workspace size. */
int rgbm_adapose_set_option(rgbm_adapose_t* h, const char* key, int value);
int rgbm_adapose_workspace_bytes(rgbm_adapose_t* h, int B, size_t* bytes);
/* img1/img2 [B,3,224,224] fp32 NCHW normalised; choose1/2 [B,1024] int32; P1/P2 [B,4,4] fp32; depths [B,24] fp32.
 * Same argument meaning as network_v5.py:418 (view1_img, view1_choose, view2_img, view2_choose, view1_proj,
 * view2_proj, depth_values).  workspace must be 256-byte aligned. */
int rgbm_adapose_forward(rgbm_adapose_t* h, int B, const float* img1, const float* img2, const int32_t* choose1,
                         const int32_t* choose2, const float* P1, const float* P2, const float* depths, void* workspace,
                         size_t workspace_bytes, const rgbm_adapose_out* out, void* stream);

/* Post-processing of one batch of network outputs into world-frame handle boxes.
 * Replaces: compute_scale_and_translation / get_3d_bbox / transform_coordinates_3d and the tail of predict()
 *   models/pose_estimator/AdaPose/lib/utils.py:40-119, models/pose_estimator/AdaPose/interface_v5.py:318-321,354-374
 * nocs1 [B,P,3] f32, depth1 [B,P] f32, r1 [B,3,3] f32, choose1 [B,P] i32, Kcrop [B,3,3] f64 (cropped intrinsics),
 * E1 [B,4,4] f64 (world->camera of view 1)  ->  bbox_world [B,8,3] f64, ts [B,4] f64 (t xyz, scale), valid [B] i32
 * (0 where the reference would return default_bbox (+10 cube), which is what bbox_world then holds). */
int rgbm_adapose_postprocess(int B, int P, int img_size, const float* nocs1, const float* depth1, const float* r1,
                             const int32_t* choose1, const double* Kcrop, const double* E1, double* bbox_world, double* ts,
                             int32_t* valid, void* stream);
/* The same post-processing with caller-provided device scratch (rgbm_adapose_postprocess_scratch_bytes(B) bytes, 8-byte aligned,
 * contents irrelevant; 0 bytes = not needed at this batch size): small batches cut the three passes over the 523 776 point pairs of
 * the exact-median search into up to 32 slices per pose (one workgroup each) instead of running one workgroup per pose — 1.1 ms of
 * a 2.9 ms B = 1 call otherwise.  Results are bit-identical to rgbm_adapose_postprocess. */
int rgbm_adapose_postprocess_scratch_bytes(int B, size_t* bytes);
int rgbm_adapose_postprocess_ws(int B, int P, int img_size, const float* nocs1, const float* depth1, const float* r1,
                                const int32_t* choose1, const double* Kcrop, const double* E1, double* bbox_world, double* ts,
                                int32_t* valid, void* scratch, size_t scratch_bytes, void* stream);

/* The `direct_regression: False`, `use_depth: True` tail of predict: back-projected predicted depth vs
 * predicted NOCS, 128-hypothesis Umeyama RANSAC, final fit over the inliers, bbox to the world frame.
 * Replaces: interface_v5.py:322-339, 348-374; lib/align.py:10-104 (estimateSimilarityUmeyama / estimateSimilarityTransform)
 * nocs1 [B,P,3] f32, depth1 [B,P] f32, choose1 [B,P] i32, Kcrop [B,3,3] f64, E1 [B,4,4] f64 (device) ->
 * bbox_world [B,8,3] f64, srt [B,13] f64 (scale, R row-major, t; scale NaN where the reference returns None), valid [B].
 * 5 <= P <= 1024.  The 5-point samples are a seeded hash (mix32(seed, 128 b + i, k) mod P) instead of np.random.randint.
 * The third branch (`use_depth: False`, cv2.solvePnPRansac) is not provided. */
int rgbm_adapose_postprocess_ransac(int B, int P, int img_size, uint32_t seed, const float* nocs1, const float* depth1,
                                    const int32_t* choose1, const double* Kcrop, const double* E1, double* bbox_world, double* srt,
                                    int32_t* valid, void* stream);
/* Tail of AdaPoseEstimator_v5.predict for `direct_regression: False`, `use_depth: False` (interface_v5.py:340-346, lib/utils.py:121-195,
 * lib/align.py:104-115): mutual NOCS matches of the two views -> epipolar gate -> DLT triangulation -> scale (median of pair ratios)
 * -> EPnP-RANSAC (100 five-point hypotheses, reprojection error 3 px) on (nocs1 * scale, pixels of view 1) -> EPnP over the inliers
 * -> VVS refinement over all points -> bbox in the world frame.  The OpenCV routines the reference calls are restated from their
 * published algorithms; RANSAC subsets come from a seeded hash (pose b, iteration i, draw k: mix32(seed, 128 b + i, k) mod P, first
 * five distinct).  nocs [B,P,3] f32; pts2d [B,P,2] f32 pixel coordinates in the ORIGINAL frame (prepare_model_input's view_pts2d);
 * K [B,3,3] f64 original intrinsics; E1 / E2 [B,4,4] f64 world -> camera.  bbox_out [B,8,3] f64 (default bbox when no match / no
 * consensus / non-finite); srt_out [B,13] f64 = scale, R (9), t (3); info_out [B,4] i32 = matches, RANSAC ok, inliers, hypotheses
 * examined; valid_out [B] i32. */
int rgbm_adapose_postprocess_pnp(int B, int P, uint32_t seed, const float* nocs1, const float* pts2d1, const float* nocs2,
                                 const float* pts2d2, const double* K, const double* E1, const double* E2, double* bbox_out,
                                 double* srt_out, int32_t* info_out, int32_t* valid_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Batched device-side input preparation.
 * Replaces: AdaPoseEstimator_v5.prepare_model_input   models/pose_estimator/AdaPose/interface_v5.py:58-170
 *           get_bbox                                  models/pose_estimator/AdaPose/lib/utils.py:10-38
 * rgb [N,H,W,3] f32 in [0,1] (the env's camera frames), mask [N,H,W] u8 (0/1), K [N,3,3] f64 camera intrinsics  ->
 * img [N,3,S,S] f32 (cropped, INTER_LINEAR-resized, ImageNet-normalised), choose [N,P] i32 (pixel indices of the
 * INTER_NEAREST-resized mask: every nonzero pixel in order, wrap-padded to P, or an ordered random P-subset),
 * pts2d [N,P,2] f32 (optional, may be NULL), Kcrop [N,3,3] f64 (crop-adjusted intrinsics), window [N,4] i32
 * (rmin,rmax,cmin,cmax), valid [N] i32 (0 where the reference returns None: empty mask / empty resized mask; such
 * frames get finite dummy outputs).  scratch: N*S*S bytes.  The P-subset is drawn by a seeded hash instead of the
 * reference's global np.random.shuffle (same distribution, reproducible).  H, W >= 440 (the reference's crop window is up to 440 pixels wide, interface_v5.py:63-88); S*S <= 65536.
 * ---------------------------------------------------------------------------------------------------------- */
int rgbm_prepare_inputs(const float* rgb_dev, const uint8_t* mask_dev, const double* K_dev, int N, int H, int W, int S, int P,
                        uint32_t seed, float* img_out, int32_t* choose_out, float* pts2d_out, double* Kcrop_out,
                        int32_t* window_out, int32_t* valid_out, uint8_t* scratch, void* stream);

/* The same, reading frame f's image and mask from entry frame_map[f] of rgb_dev / mask_dev (e.g. the controller's view
 * queue [T*N,H,W,...]) instead of gathering the selected views first; frame_map[f] < 0 = no such view
 * (treated as an empty mask: valid 0).  K_dev, every output and the subset hash stay indexed by f. */
int rgbm_prepare_inputs_indexed(const float* rgb_dev, const uint8_t* mask_dev, const double* K_dev, const int32_t* frame_map_dev,
                                int N, int H, int W, int S, int P, uint32_t seed, float* img_out, int32_t* choose_out,
                                float* pts2d_out, double* Kcrop_out, int32_t* window_out, int32_t* valid_out, uint8_t* scratch,
                                void* stream);
/* Both of the above with a hash offset: frame f of this call draws its 1024-subset as frame frame0 + f would in one call over the
 * whole batch, so a batch prepared in pieces (estimate() uploads and prepares host frames chunk by chunk while the previous chunk
 * computes) chooses the same pixels as the unchunked call.  frame_map_dev may be null (frames in order). */
int rgbm_prepare_inputs_ex(const float* rgb_dev, const uint8_t* mask_dev, const double* K_dev, const int32_t* frame_map_dev,
                           int frame0, int N, int H, int W, int S, int P, uint32_t seed, float* img_out, int32_t* choose_out,
                           float* pts2d_out, double* Kcrop_out, int32_t* window_out, int32_t* valid_out, uint8_t* scratch,
                           void* stream);

/* Per-env mask extent for the controller's view queue.
 * Replaces: the np.nonzero / np.where loop of ControlInterface.add_view   models/controller/rl_pose.py:130-149
 * mask [N,H,W] u8 -> ext [N,4] i32 = (row min, col min, row max, col max), (2H, 2W, 0, 0) for an empty mask; count [N]. */
/* Projection matrices the network takes (interface_v5.py:264-270): P = eye(4), P[:3, :] = Kcrop @ E[:3, :] in fp64, stored as
 * float32.  Kcrop_dev [N][3][3] f64 (rgbm_prepare_inputs' Kcrop_out), E_dev [N][4][4] f64, P_dev [N][4][4] f32. */
int rgbm_projection(const double* Kcrop_dev, const double* E_dev, float* P_dev, int N, void* stream);
int rgbm_mask_extent(const uint8_t* mask_dev, int N, int H, int W, int32_t* ext_out, int32_t* count_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Controller step on the device: everything ControlInterface.step does between the simulator and the
 * estimator, one thread per environment, float64 in numpy's evaluation order.
 * ---------------------------------------------------------------------------------------------------------- */
/* Replaces: utils.transform.lookat_quat   utils/transform.py:50-99 (per-row branches; batch_zero = its whole-batch norm test)
 * dir [N,3] f64 -> quat [N,4] f64 (w,x,y,z).  The reference takes an eigenvector of Horn's matrix whose SIGN is LAPACK's
 * choice; here the first non-negligible component is positive (same rotation). */
int rgbm_lookat_quat(const double* dir_dev, int N, int batch_zero, double* quat_dev, void* stream);
/* Replaces: the action decode of ControlInterface.step   models/controller/rl_pose.py:390-408 (action_type "pose")
 * action [N,lda] f32 (device) -> pose [N,7] f64: clip(a[:3] + pose_mid, pose_min, pose_max), lookat_quat((1, dy, dz)).
 * pose_mid / pose_min / pose_max: 3 host doubles each. */
int rgbm_control_action_to_pose(const float* action_dev, int lda, const double* pose_mid, const double* pose_min,
                                const double* pose_max, int N, double* pose_dev, void* stream);
/* Replaces: ControlInterface.get_reward   models/controller/rl_pose.py:225-358, including quat_to_axis' batch scramble
 * (utils/transform.py:234) and LOSS:far aliasing the scaled far term (:247, :322).  All pointers are device pointers. */
typedef struct rgbm_control_reward_args {
  const float* action;        /* [N,lda] f32 policy action: xyz, dy, dz, -, view weights (T) */
  const double* cam_pose;     /* [N,7]   env.camera_pose(robot_frame=True) */
  const double* target;       /* [N,7]   last_pose_target */
  const float* move_success;  /* [N]     cam_move_to()[0] as float32 */
  const double* bbox;         /* [N,4]   bbox_queue[s % T] */
  const double* avail;        /* [N]     available[s % T] */
  const double* gt_bbox;      /* [N,8,3] gt_bbox[s] */
  const double* pred_bbox;    /* [N,8,3] pred_bbox[s] */
  const double* pose_cur;     /* [N,7]   pose_queue[s] */
  const double* pose_prev;    /* [N,7]   pose_queue[s-1] */
  const double* robot_pose;   /* [N,7]   env.robot_pose() */
  const double* success;      /* [N] */
  double* reward;             /* [N] out */
  double* terms;              /* [17][N] out or NULL: the 14 REW:* terms in sum order, LOSS:center_diff, LOSS:open_diff, LOSS:far */
  double coef[14];            /* diff, move_success, move_period, far, ori, xyz_lookat, bbox, bbox_boundary, have_bbox, center,
                                 open, view, view_norm, success */
  double proper_pos[3];
  double precision2;          /* precision^2: 0.01 for mugs, 0.04 otherwise */
  int N, T, lda, pots, first; /* T = max_steps (view weights per action); first = (accumulate_steps == 0) */
  int pad_;
} rgbm_control_reward_args;
int rgbm_control_reward(const rgbm_control_reward_args* args, void* stream);
/* Replaces: the centre / axis math of ControlInterface.call_manipulation   models/controller/rl_pose.py:364-377
 * est [N,8,3] f64 -> center [N,3], direction [N,3,3]. */
int rgbm_control_grasp_frame(const double* est_dev, int N, double* center_dev, double* direction_dev, void* stream);

/* Synthetic camera of the MultiVecEnv stand-in (no reference code: the reference renders with SAPIEN).
 * Produces what MultiVecEnv.get_image() returns (env/my_vec_env.py:266, base_manipulation.py:653-687) for a scene of one
 * oriented box per env.  rays [N,12] is scratch written by rgbm_synth_camera and read by rgbm_synth_render.
 * Bit-identical to oracle/synth_env_ref.py. */
typedef struct rgbm_synth_scene {
  const double* cam_pose;     /* [N,7] camera pose in the robot frame (x forward, y left, z up) */
  const double* robot_pose;   /* [N,7] robot root (position used) */
  const double* box;          /* [N,15] centre (3), axis rows X,Y,Z (9), half extents (3) */
  double fx, fy, cx, cy;
  int N, H, W, env0;
} rgbm_synth_scene;
int rgbm_synth_camera(const rgbm_synth_scene* scene, double* K_dev, double* E_dev, double* rays_dev, void* stream);
int rgbm_synth_render(const rgbm_synth_scene* scene, const double* rays_dev, float* color_dev, uint8_t* mask_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * PPO rollout storage.
 * Replaces: RolloutStorage.compute_returns   algo/ppo/ppo/storage.py:50-64
 * rewards/values/returns/adv [T,N] f32, dones [T,N] u8, last_values [N] f32.
 * sums: device f64 buffer with room for 2 + 2*ceil(N/256) doubles; on return sums[0..1] = {sum(adv), sum(adv^2)}
 * of the un-normalised advantages (all-reduce these two over ranks for the global normalisation), then
 * rgbm_adv_normalise applies (adv - mean) / (unbiased_std + 1e-8) with count_total = T*N summed over ranks.
 * ---------------------------------------------------------------------------------------------------------- */
int rgbm_gae(int T, int N, const float* rewards, const uint8_t* dones, const float* values, const float* last_values,
             float gamma, float lam, float* returns, float* adv, double* sums, void* stream);
int rgbm_adv_normalise(int64_t n_local, float* adv, const double* sums, double count_total, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * PPO policy (actor-critic MLPs obs -> h0 -> h1 -> h2 -> {act, 1}, ELU) on a flat fp32 parameter vector laid out in the
 * reference's state_dict order (log_std, actor.{0,2,4,6}.{weight,bias}, critic.{0,2,4,6}.{weight,bias}).
 * Replaces: ActorCritic.act / act_inference / evaluate   algo/ppo/ppo/module.py:73-107
 *           one minibatch of PPO.update                   algo/ppo/ppo/ppo.py:472-528
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct rgbm_policy_layout {
  int dims[5];        /* obs, h0, h1, h2, act */
  int log_std;        /* offsets (in floats) into the flat parameter vector */
  int w[2][4];        /* [net: 0 actor, 1 critic][layer] weight offsets, row-major [out][in] */
  int b[2][4];
  int total;          /* number of parameters */
} rgbm_policy_layout;
/* mode 0: act (actions = mu + exp(2*log_std)*noise, log-prob, value, mu); 1: act_inference (mu only);
 * 2: evaluate (log-prob of the given actions, value, mu).  obs [n,dims[0]], noise/actions/mu [n,dims[4]], logp/value [n]. */
int rgbm_policy_forward(const float* params, const rgbm_policy_layout* L, int n, int mode, const float* obs, const float* noise,
                        float* actions, float* logp, float* value, float* mu, void* stream);
/* scratch floats needed by rgbm_ppo_minibatch_fwd_bwd for n rows */
int rgbm_ppo_partial_floats(const rgbm_policy_layout* L, int n, size_t* count);
/* forward + clipped-surrogate / clipped-value / entropy loss + backward for one minibatch of n rows.
 * grads_flat [total+4]: d loss / d params (mean over the n rows), then {sum surrogate, sum value loss, sum KL, rows}.
 * With several ranks: all-reduce(sum) grads_flat, then call rgbm_ppo_clip_adam with inv_world = 1/world. */
int rgbm_ppo_minibatch_fwd_bwd(const float* params, const rgbm_policy_layout* L, int n, const float* obs, const float* actions,
                               const float* old_logp, const float* adv, const float* returns, const float* old_values,
                               const float* old_mu, const float* old_log_std, float clip, float vcoef, float ecoef,
                               float* partial_scratch, float* grads_flat, void* stream);
/* clip_grad_norm_(max_norm) + KL-adaptive learning rate (ppo.py:486-495) + Adam(0.9,0.999,1e-8) in one launch.
 * opt_state: 48-byte device record {int t, n_updates; float lr, last_kl, last_norm, pad; double sum_surr, sum_vloss}. */
int rgbm_ppo_clip_adam(float* params, const float* grads_flat, float* exp_avg, float* exp_avg_sq, void* opt_state,
                       const rgbm_policy_layout* L, float inv_world, float max_norm, float desired_kl, float lr_min,
                       float lr_max, int adaptive, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Layer-level entry points (used by the parity tests and by the Python host for pieces it drives itself).
 * ---------------------------------------------------------------------------------------------------------- */
/* Generic N-D convolution on channels-last tensors, weights given in PyTorch layout on the host
 * (nn.Conv2d/Conv3d: [Cout][Cin][KD][KH][KW]; nn.ConvTranspose3d(k3,s2,p1,op1) when transposed=1: [Cin][Cout][3][3][3]).
 * in_dev [N][D][H][W][Cin_pad], out_dev [N][Do][Ho][Wo][Cout_pad] in `dtype`; act 0 none 1 relu 2 prelu 3 tanh;
 * res_mode 0 none 1 add before activation 2 add after activation. */
int rgbm_conv_nd(int dtype, const void* in_dev, int N, int D, int H, int W, int Cin, int Cin_pad, const float* w_host,
                 int Cout, int Cout_pad, int KD, int KH, int KW, int stride_d, int stride_hw, int pad_d, int pad_hw,
                 int dil_hw, int transposed, const float* bias_host, const float* bn_scale_host, const float* bn_shift_host,
                 const void* res_dev, int res_mode, int act, float slope, void* out_dev, void* stream);
/* Halo-tiled 3-D conv of the cost-regularisation stack, one layer: layer 0..6 = conv0..conv6 (k3, pad 1, stride 1/2),
 * 7..9 = conv7/conv9/conv11 (ConvTranspose3d k3 s2 p1 op1).  Channels are fixed by the layer (network_v5.py:263-278);
 * in_dev [N][D][H][W][Cin], out_dev [N][Do][Ho][Wo][Cout], folded BN scale/shift on the host, ReLU, optional
 * post-activation residual res_dev (same layout as out). */
int rgbm_conv3d_tile(int layer, int dtype, const void* in_dev, int N, int D, int H, int W, const float* w_host,
                     const float* bn_scale_host, const float* bn_shift_host, const void* res_dev, void* out_dev, void* stream);
/* PSPUpsample (pspnet.py:100-107: x2 bilinear align_corners=True -> Conv2d 3x3 pad 1 + bias -> activation) as the network runs
 * it: a 1x1 GEMM at the low resolution with the nine taps stacked on the output channels (z_scratch_dev: V*h*w*9*Cout elements
 * of `dtype`), then the tap-combining kernel.  in_dev [V][h][w][Cin], w_host [Cout][Cin][3][3], out_dev [V][2h][2w][Cout]. */
int rgbm_upsample_conv3x3(int dtype, const void* in_dev, int V, int h, int w, int Cin, const float* w_host, int Cout,
                          const float* bias_host, int act, float slope, void* z_scratch_dev, void* out_dev, void* stream);
/* The PSPNet tail as the 16-bit / split-pair network runs it: up_3 (PSPUpsample 64 -> 64, PReLU `slope`) and `final`
 * (pspnet.py:136, Conv2d 1x1 64 -> 32 + bias) in one kernel (upconv_final.hip).  in_dev [V][h][w][64] (h, w multiples of 8),
 * w3_host [64][64][3][3], b3_host [64], wf_host [32][64], bf_host [32]; dtype RGBM_BF16 / RGBM_F16 / RGBM_BF16X3.
 * out_f32 is an ENUM since round 5 (it was a boolean before; any other value is refused with RGBM_ERR_ARG):
 *   0 = out_dev [V][2h][2w][32] in `dtype`;
 *   1 = plain fp32 output (RGBM_BF16X3 only: the feature map the split-pair plane sweep gathers from);
 *   2 = IEEE f16 output through the packed-f16 tail (RGBM_BF16 only: the bf16 network's `sweep_f16` feature map; refused when
 *       `final`'s weights do not fit f16 — a value >= 65504 or more than 0.1 % of the weight mass below 2^-14). */
int rgbm_upsample_conv3x3_final(int dtype, const void* in_dev, int V, int h, int w, const float* w3_host, const float* b3_host,
                                float slope, const float* wf_host, const float* bf_host, void* out_dev, int out_f32, void* stream);
/* The ResNet stem as the 16-bit / split-pair network runs it (stem.hip): Conv2d 7x7 stride 2 pad 3 (3 -> 64, no bias, pspnet.py:37)
 * -> ReLU -> MaxPool2d 3x3 stride 2 pad 1 (pspnet.py:39) in one kernel.  img1_dev / img2_dev [B][3][S][S] fp32 NCHW (views
 * 0..B-1 / B..2B-1), w_host [64][3][7][7]; out_dev [2B][S/4][S/4][64] in `dtype` (RGBM_BF16 / RGBM_F16 / RGBM_BF16X3); S % 32 == 0. */
int rgbm_stem(int dtype, const float* img1_dev, const float* img2_dev, const float* w_host, void* out_dev, int B, int S, void* stream);
int rgbm_maxpool3x3s2(int dtype, const void* in_dev, void* out_dev, int V, int H, int W, int C, void* stream);
int rgbm_resize_bilinear_ac(int dtype, const void* in_dev, void* out_dev, int V, int Hs, int Ws, int C, int Ho, int Wo,
                            void* stream);
int rgbm_adaptive_avgpool(int dtype, const void* in_dev, void* out_dev, int V, int H, int W, int C, int S, void* stream);
/* fused plane-sweep volume for views [0,V): feat [V][H][W][32] (dtype), P [V][4][4] f32 (views ordered side*B+b),
 * depths [B][D] f32 -> vol [V][D][H][W][32]; homog_scratch: V*12 floats */
int rgbm_build_volume(int dtype, const void* feat_dev, const float* P_views_dev, const float* depths_dev, float* homog_scratch,
                      void* vol_dev, int V, int B, int D, int H, int W, void* stream);
/* conv0 of the cost-regularisation net with the plane sweep fused in, depth-sweeping kernel (bf16 only; what
 * cost_impl = 3 runs): feat [V][H][W][32] bf16 (views 0..B-1 = view 1, B..2B-1 = view 2), P_views [V][4][4], depths [B][D],
 * homog_scratch [V*12] floats, w_host [8][32][3][3][3] + folded BatchNorm scale/shift [8] on the host ->
 * out [V][D][H][W][8] bf16 = ReLU(BN(conv3d(feat[v] + homo_warping(feat[partner(v)])))) (network_v5.py:262,378-430). */
int rgbm_conv0_sweep(const void* feat_dev, const float* P_views_dev, const float* depths_dev, float* homog_scratch,
                     const float* w_host, const float* bn_scale_host, const float* bn_shift_host, void* out_dev,
                     int V, int B, int D, int H, int W, void* stream);
/* the same kernel for either 16-bit storage type (dtype = RGBM_BF16 or RGBM_F16; feat / out in that type); dtype = RGBM_BF16X3
 * runs conv0_sweep_x3_kernel: feat is PLAIN fp32 [V][H][W][32], out is the split-pair tensor [V][D][H][W][8] */
int rgbm_conv0_sweep_dt(int dtype, const void* feat_dev, const float* P_views_dev, const float* depths_dev, float* homog_scratch,
                        const float* w_host, const float* bn_scale_host, const float* bn_shift_host, void* out_dev,
                        int V, int B, int D, int H, int W, void* stream);
/* what a bf16 net runs by default since round 5 (option "sweep_f16" = 1): feat is f16 [V][H][W][32] (written so by the net's `final`
 * layer), the conv0 weights are rounded to f16, the blend is packed-f16 arithmetic, out is bf16 [V][D][H][W][8].  Features beyond
 * +-65504 saturate (as in an fp16 net); set the option to 0 for the all-bf16 form (rgbm_conv0_sweep). */
int rgbm_conv0_sweep_f16feat(const void* feat_f16_dev, const float* P_views_dev, const float* depths_dev, float* homog_scratch,
                             const float* w_host, const float* bn_scale_host, const float* bn_shift_host, void* out_bf16_dev,
                             int V, int B, int D, int H, int W, void* stream);
/* debugging access to a named intermediate of the last rgbm_adapose_forward on (h, B, workspace):
 * converts it to fp32 into out_dev (elems = capacity in floats); *n_elems returns its size.  Intermediates of the
 * PSPNet phase are overwritten by the cost-volume phase, so pass stop_after = 1 to rgbm_adapose_forward_ex first.
 * The 3-D taps "c0" "c2" "c4" "c6" "u7" "u9" exist as whole tensors only with option sparse_dec = 0: with the default (sparse cost
 * regularisation) the call fails with an error instead of returning tensors that are stale outside the dependency cones.
 * "pf96" of a split-pair net is converted back from the hi / lo pairs the pose MLP consumed. */
int rgbm_adapose_forward_ex(rgbm_adapose_t* h, int B, const float* img1, const float* img2, const int32_t* choose1,
                            const int32_t* choose2, const float* P1, const float* P2, const float* depths, void* workspace,
                            size_t workspace_bytes, const rgbm_adapose_out* out, int stop_after, void* stream);
/* rgbm_adapose_forward replayed from a hipGraph (small batches — the reference calls the network at batch 1 per env,
 * interface_v5.py:213-227,259-280, and ships num_envs: 8 — are launch-bound: ~150 launches per forward).  The first call with a
 * given (B, every device pointer, workspace, outputs) runs the forward once eagerly, captures its launch sequence on `stream` and
 * keeps the instantiated graph (up to 8 per handle, least recently used evicted; dropped when an option changes); later calls with the
 * same arguments replay it with one hipGraphLaunch.  The caller keeps every buffer alive and at the same address.  `stream` must be
 * a created (non-null) stream.  *n_nodes (optional) = graph nodes (kernel launches + copies) of this batch size; *captured
 * (optional) = 1 when this call did the capture, 0 on a replay, -1 when it ran eagerly because rgbm_prof_start is active.
 * Pass the same `stream` for every call on one set of buffers: the captured launches of a small batch use the K-split scratch of the stream
 * they were captured on (rgbm_debug_flags 16384), which other streams' forwards do not share. */
int rgbm_adapose_forward_graph(rgbm_adapose_t* h, int B, const float* img1, const float* img2, const int32_t* choose1,
                               const int32_t* choose2, const float* P1, const float* P2, const float* depths, void* workspace,
                               size_t workspace_bytes, const rgbm_adapose_out* out, void* stream, int32_t* n_nodes, int32_t* captured);
int rgbm_adapose_graph_clear(rgbm_adapose_t* h);
int rgbm_adapose_fetch(rgbm_adapose_t* h, int B, void* workspace, const char* name, float* out_dev, size_t capacity,
                       size_t* n_elems, void* stream);

/* Live per-kernel timing for bench.py's roofline figure: between start and stop every convolution launch is
 * bracketed by HIP events recorded on its own stream.  stats: host double[RGBM_PROF_ROWS * 4], one row per kernel family,
 * columns {launches, total ms, algorithmic FLOPs, algorithmic bytes}; rgbm_prof_rows() returns RGBM_PROF_ROWS of the
 * library that is loaded (size the buffer from it when binding dynamically).  Rows:
 *  0..7   conv_igemm_glds_kernel<dtype, BCH> (row = dtype*4 + {0:16,1:32,2:64,3:128}-channel tile; dtype 0 f32, 1 16-bit)
 *  8 / 9  conv3d_tile_kernel f32 / (unused)          10 / 11  conv3d_tile_kernel conv0 + fused plane sweep f32 / 16-bit
 *  12 / 13 conv_igemm_ws_kernel 128 x 256 tile (and conv_igemm_v3_kernel) f32 / 16-bit
 *  14     conv0_sweep[_persistent]_kernel (16-bit)            15  conv_igemm_ws64_kernel (16-bit)
 *  16..25 conv3d_tile_kernel 16-bit per layer (conv0..conv6, conv7, conv9, conv11)
 *  26..29 bf16x3: generic implicit GEMM, 3-D layers, conv0 + plane sweep, ws 128 x 256 tile
 *  30     (unused since round 6)                     31 / 32  ws 256 x 128 tile 16-bit / (32: unused since round 6)
 *  33     ws 256 x 128 tile bf16x3                   34 / 35 / 36  ws 64 x 256 four-multiply-wave tile bf16x3 / 16-bit / f32
 *  37 / 38 upconv_combine_kernel 16-bit / 4-byte storage                39  upconv_final_kernel
 *  40 / 41 conv_igemm_m32_kernel<T, 128, 64 / 128 / 256>: the 128-pixel tail and small-batch launches of rows 31 / 33 (16-bit / bf16x3;
 *          rows 31 / 33 are conv_igemm_m32_kernel<T, 256, 256> under the default gemm_kernel = 2, conv_igemm_ws_kernel<T, wide> under gemm_kernel = 0)
 * stop synchronises on the recorded events. */
#define RGBM_PROF_ROWS 42
/* A/B switches for kernel benchmarking and the parity tests of the non-default kernel variants (0 = normal operation; bits OR together):
 *      8  no persistent ws kernels (generic tiles)
 *     16  treat every conv as non-uniform taps (v3 / generic kernels)                          64  v3 kernel instead of the ws kernel
 *    128  no ws64 kernel      256  ws64 without the row-halo variant      512  generic resize instead of the x2 kernel
 *  32768  layer2's 128-channel layers at one to four poses on the 64 x 256 tile of conv_igemm_ws_kernel as before (default since round 6:
 *          64-channel x 128-pixel tiles of conv_igemm_m32_kernel with the K split, residual added in the epilogue)
 *  16384  no K split of the 256-channel GEMM's launches of few tiles (small batches; default since round 6: the K loop of a 64 / 128-channel x
 *          128-pixel tile is cut into up to four parts, one workgroup each, when the tiles fill at most half the CUs; fixed-order fp32 sum)
 *   2048  per-point NOCS branch as in rounds 1-5: a gather launch and six fp32 1x1 GEMM launches (default since round 6: point_mlp_kernel)
 *   1024  PSP stage as in rounds 1-5: copy, pooling, four 1x1 GEMM launches, four resize launches (default since round 6: one concat launch;
 *          up to 32 views also pooling + the four 1x1 convs in one launch on the vector pipe)
 *   4096  halo-tile conv0 instead of the plane-sweep kernels             65536  128 x 256 ws tile even where the 256-channel tiles apply
 *          (4 / 8192 / 131072 selected the register-staged kernel, the 256 x 256 two-group kernel and the row-halo wide tile of rounds
 *          1-2: measured slower, removed in round 6; the bits are ignored)
 * 262144  generic tile instead of the 64 x 256 four-wave ws tile         1048576  ws request waves walk K taps outer, channel blocks inner
 *                                                                                  (default since round 3: channel block outer, taps inner)
 * 2097152  plane sweep of a bf16 feature map (option sweep_f16 = 0): blend on fp32 FMAs from inline asm; fp16 nets: the packed-f16 blend
 *          instead of their fp32-accumulating one                        4194304  bf16 feature map: v_perm + v_dot2_f32_bf16 (8-bit bilinear
 *                                                                                  weights; the round-4 default; since round 5: plain fp32 blend)
 * 8388608  one-workgroup-per-pose post-processing even with scratch      16777216  no 64 x 256 tile for small persistent launches
 * 33554432 post-processing: generic fp64 radix selection of the median only (the fall-back of the default selection on fp32
 *          approximations; same result bit for bit)                    67108864  post-processing: guard band in every even-count pose
 * 134217728 implicit-GEMM request waves: 64-bit global addresses + zero page instead of buffer descriptors (the form before round 5)
 * 268435456 plane sweep of an f16 feature map (bf16 nets, sweep_f16 = 1; fp16 nets since round 6): one workgroup per tile instead of the
 *          persistent kernel            536870912  sparse tail: the round-3..5 point kernel instead of prob_sparse2_kernel
 * 1073741824 256-channel GEMM: tail and small launches on 256-channel x 128-pixel tiles / the 64 x 256 ws tile as in round 5 */
int rgbm_debug_flags(int flags);
/* Kernel choice - tile shape, and with it the order of the fp32 sums - depends on a launch's GEMM rows, i.e. on the batch size: the
 * same pose in batches of different sizes agrees to the storage type's rounding (1e-6 .. 1e-5 relative in fp32 / bf16x3), not bit for
 * bit.  Equal-sized batches (and the two half batches of split_streams) are bit-identical.
 * dispatch thresholds (process-wide, like the debug flags).  "ws_min_rows": GEMM rows (output pixels of a conv launch) from which
 * the persistent role-specialised implicit-GEMM kernels are used instead of the generic tiles; 0 (default) = 1024 (measured at
 * B = 1 .. 8 in every storage type; rounds 1-3 used 65536).  Launches whose 64-channel x 256-pixel tiles fit one round of the
 * persistent grid take that tile shape (debug flag 16777216 disables it).
 * "gemm_kernel": kernel of the launches whose output channels are a multiple of 256 (layer3 / layer4 / up_1) in the 16-bit storage
 * types: 0 = conv_igemm_ws_kernel (16x16x32 MFMAs, 256 x 128 tile), 1 = conv_igemm_m32_kernel on 256 x 128 tiles (32x32x16 MFMAs,
 * one multiply wave per SIMD), 2 (default) = whole rounds of 256 x 256 tiles + a 256 x 128 tail launch.  1 and 2 add a pre-activation
 * residual on the matrix pipe (identity K steps) and give bit-identical results; 0 differs from them in the last bit of a few
 * residual sums per million.  Debug flag 134217728: the persistent kernels' request waves use 64-bit global addresses and a zero page
 * instead of buffer descriptors (A/B). */
int rgbm_set_tuning(const char* key, long long value);
/* Achievable-peak probes (the peaks the box reaches, next to the datasheet ones; no reference counterpart).  Asynchronous:
 * the caller times them with events on `stream`.  rgbm_microbench_mfma: a bare v_mfma_f32_16x16x32_bf16 stream, 8 waves per CU on
 * every CU, `iters` x 32 MFMAs per wave on constant (random_operands = 0) or pseudo-random operands; `scratch` holds
 * rgbm_microbench_mfma_scratch_floats floats; *flops = the flops the launch executes.  rgbm_microbench_copy: grid-stride 16-byte copy. */
int rgbm_microbench_mfma_scratch_floats(int* n);
int rgbm_microbench_mfma(float* scratch, int iters, int random_operands, double* flops, void* stream);
int rgbm_microbench_copy(const void* src, void* dst, size_t bytes, void* stream);
int rgbm_prof_rows(void);
int rgbm_prof_start(void);
int rgbm_prof_stop(double* stats);
/* Restrict the per-launch events to one row of the table (-1 = every row, the default): two events per launch cost ~3.5 us, i.e.
 * 0.35 ms per 256-pose forward when every launch is bracketed; bench.py brackets only the dominant kernel inside its timed region. */
int rgbm_prof_select(int row);

#ifdef __cplusplus
}
#endif
#endif /* RGBM_H_ */
