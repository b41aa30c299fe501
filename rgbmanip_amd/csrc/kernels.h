// Host-side launchers of the helper kernels (misc_kernels.hip, head_kernels.hip, postproc.hip, ppo_kernels.hip).
#pragma once
#include <vector>

#include "common.h"

namespace rgbm {

// misc_kernels.hip
int launch_nchw_to_nhwc_pad(int dtype, const float* in, void* out, int V, int C, int H, int W, int Cp, hipStream_t s);
int launch_maxpool3x3s2(int dtype, const void* in, void* out, int V, int H, int W, int C, hipStream_t s);
int launch_resize_bilinear_ac(int dtype, const void* in, void* out, int V, int Hs, int Ws, int C, int Ho, int Wo, int ldo,
                              int ch_off, hipStream_t s);
int launch_copy_channels(int dtype, const void* in, void* out, long long npix, int C, int ldo, int ch_off, hipStream_t s);
int launch_adaptive_avgpool(int dtype, const void* in, void* out, int V, int H, int W, int C, int S, hipStream_t s);
int launch_adaptive_avgpool_multi(int dtype, const void* in, void* const* outs, const int* bins, int nb, int V, int H, int W, int C,
                                  hipStream_t s);
// PSP stage of small batches (misc_kernels.hip): pooling + the four 512 -> 128 1x1 convs in one launch (small batches); the whole concat (copy + four resizes) in one
int launch_psp_pool_conv(int dtype, const void* in, int ldi, const void* const* w, void* const* outs, const int* bins, int V, int H, int W,
                         int act, float slope, hipStream_t s);
int launch_psp_resize_cat(int dtype, const void* f, const void* const* stages, const int* bins, void* out, int V, int Ho, int Wo, hipStream_t s);
int launch_homography(const float* P_views, float* out, int V, int B, hipStream_t s);
int launch_build_volume(int dtype, const void* feat, const float* homog, const float* depths, void* vol, int v0, int Vc, int V,
                        int B, int D, int H, int W, hipStream_t s);

// upconv.hip — tap-combining half of PSPUpsample evaluated as a low-resolution 1x1 GEMM (layers.h, UpConvLayer):
// z [V][h][w][9*Co] (tap-major) -> out [V][2h][2w][ldo] = act(bias + sum_t bilinear(z_t)(p + t))
int launch_upconv_combine(int dtype, const void* z, const float* bias, void* out, int V, int h, int w, int Co, int ldo, int act,
                          float slope, hipStream_t s);
// up_3 + final of the PSPNet tail in one kernel (upconv_final.hip): x [V][h][w][64] -> out [V][2h][2w][32]
int launch_upconv_final(int dtype, const void* x, const void* wz, const float* bias, float slope, const void* wf, const float* biasf,
                        void* out, int out_f32, int V, int h, int w, hipStream_t s, const void* wf_f16 = nullptr);      // wf_f16: `final` weights in f16 (out_f32 == 2)

// bn_kernels.hip — per-sample (train-mode, batch 1) BatchNorm3d + ReLU + skip add, in place on the un-normalised conv output
size_t bn_scratch_bytes(int V);
int launch_bn_per_sample(int dtype, void* y, const void* res, const float* gamma, const float* beta, void* scratch, int V,
                         long long nvox, int C, int relu, hipStream_t s);

int launch_to_f32(int dtype, const void* in, float* out, long long n, hipStream_t s);

// head_kernels.hip
int launch_gather_points(int dtype, const void* feat, const int* choose, float* out, int V, int P, int HW, int C, hipStream_t s);
int launch_prob_softmax_depth(int dtype, const void* u11, const float* wprob, const int* choose, const float* depths,
                              float* prob, float* depth_out, int v0, int Vc, int B, int P, int D, int H, int W, int classmajor,
                              hipStream_t s);
int launch_fuse_points(int dtype, const void* feat, const float* homog, const float* depths, const int* choose,
                       const float* prob, float* out, int V, int B, int P, int D, int H, int W, int ldo, int ch_off,
                       hipStream_t s, int Vn = -1);      // Vn: views 0 .. Vn - 1 only (default: all V)
int launch_mean_points(int dtype, const void* in, float* out, float* scratch, int V, int P, int C, hipStream_t s);      // dtype of `in`: F32 or F16
int launch_f32_to_f16(const float* in, void* out, long long n, hipStream_t s);
// ResNet stem in one kernel (stem.hip): NCHW fp32 images -> maxpool3x3s2(relu(conv7x7s2)) [V][S/4][S/4][64]
void stem_pack(const float* w, std::vector<float>& packed);
int launch_stem(int dtype, const float* img1, const float* img2, const void* wpk, void* out, int B, int V, int S, hipStream_t s);
int launch_f32_to_bx3(const float* in, void* out, long long n, hipStream_t s);     // plain fp32 -> split pairs (n % 4 == 0; in place allowed)
int launch_view_linear(const float* x, const float* W, const float* bias, float* out, int V, int I, int O, int ldw, int i0,
                       int relu, hipStream_t s);
// head_kernels.hip: gather + the six per-point layers of the NOCS branch in one launch
struct PointMlpDesc {
  const float* table;      // device: point_mlp_table_floats() floats from point_mlp_pack (weights in MFMA-fragment order, then biases)
  const void* feat;        // [V][HW][32] in the feature map's storage type
  const int* choose;       // [V * P]
  float* nocs4;            // [V * P][4]
  float* pf;               // channel 0 of the 64 output channels, row stride ldpf
  int ldpf, P, HW;
  long long N;             // V * P, a multiple of 64
};
int point_mlp_table_floats();
void point_mlp_pack(const float* const w[6], const float* const b[6], float* table);      // host; w[l]: [Cout][Cin] row-major fp32 of instance_color.0, nocs_head.0/2/4, nocs_pts_mlp.0/2
int launch_point_mlp(int feat_dtype, const PointMlpDesc& d, hipStream_t s);
int launch_ortho6d(const float* r6, float* R, int V, hipStream_t s);
// consumers that finish a mean over points themselves (one launch less per mean): launch_mean_points_partial writes the slices' sums
int launch_mean_points_partial(int dtype, const void* in, float* scratch, int V, int P, int C, hipStream_t s);
int launch_view_linear_mean(const float* partial, int P, float* mean_out, const float* W, const float* bias, float* out, int V, int I, int O,
                            int ldw, int i0, int relu, hipStream_t s);
int launch_pose_heads_mean(const float* partial, int P, float* mean_out, float* R, float* const w[3][3], float* const b[3][3],
                           float* const out[3], const int odim[3], int V, hipStream_t s);
// the three regression heads (3 x Linear + ReLU each) of every view in one launch; out[h] is [V][odim[h]]
int launch_pose_heads(const float* pf2, float* const w[3][3], float* const b[3][3], float* const out[3], const int odim[3], int V,
                      hipStream_t s);
// input / output staging of AdaPose::forward in one launch each
int launch_stage_in(const float* P1, const float* P2, const int* c1, const int* c2, float* Pviews, int* choose, int B, int P, hipStream_t s);
int launch_stage_out(const float* nocs4, const float* depth, const float* R, const float* tv, const float* sv, float* const nocs[2],
                     float* const dep[2], float* const r[2], float* const t[2], float* const sz[2], int B, int P, int view2, hipStream_t s);
int launch_copy_cols(const float* in, float* out, long long rows, int ldi, int ldo, int n, hipStream_t s);

// conv3d_tile.hip — halo-tiled 3-D conv for the cost-regularisation stack
struct Conv3dTileDesc {
  const void* in; const void* wgt; void* out; const float* bias; const void* res;
  int N, Di, Hi, Wi;            // input tensor [N][Di][Hi][Wi][CIN]
  int Do, Ho, Wo;               // output tensor dims
  int Dq, Hq, Wq;               // tile-enumeration grid (= output dims; = input dims for transposed)
  int ntd, nth, ntw;            // filled by the launcher
  int Cout, relu;
  const void* feat; const float* homog; const float* depths; int v0, V, B;   // fused-warp mode only
  int prof_variant; double algo_flops, algo_bytes;
  int out_classmajor;           // transposed only: write each sub-pixel class as its own dense [N][Dq][Hq][Wq][C] volume
  int feat_f16;      // depth-sweeping conv0 of a bf16 net: `feat` and `wgt` are f16 (AdaPose::feat_f16()), the output stays bf16
  const int* tile_list; const int* tile_count;      // depth-sweeping conv0 only: walk tile_list[0 .. tile_count[0]) instead of all tiles
  int tile_mask_stride;             // bytes between consecutive views' masks (0: nth * ntw)
  const unsigned char* tile_mask;   // optional [N][nth][ntw]: a workgroup whose (view, row tile, column tile) byte is 0 returns at once (its
                                    // output tile is never read: sparse decoder, see launch_decoder_tile_masks); all depth tiles share a byte
};
extern int g_debug_flags;
extern long long g_ws_min_rows;
extern int g_tuning_version;
extern int g_gemm_kernel;
void conv3d_tile_pack(const float* w, const float* scale, int Cin, int Cout, int coutp, bool transposed, int dtype,
                      std::vector<float>& packed);
int launch_conv3d_tile(int layer, int dtype, const Conv3dTileDesc& d, hipStream_t s);
// prob_sparse.hip: sparse cost regularisation — tile masks of every 3-D layer from the chosen pixels' dependency cones, and the
// compacted tile list of the depth-sweeping conv0 (layer ids as launch_conv3d_tile's; 0 = the sweep)
int sparse_mask_bytes_per_view(int S);
int sparse_mask_offset(int S, int layer);
int launch_sparse_masks(const int* choose, int v0, int Vc, int P, int S, unsigned char* masks, int* sweep_list, int* sweep_count,
                        hipStream_t s);
// tile grid of a layer's halo-tile kernel in the given storage type (for sizing Conv3dTileDesc::tile_mask)
int conv3d_tile_dims(int layer, int dtype, int* TD, int* TH, int* TW);

// conv0_sweep.hip — conv0 + fused plane sweep, depth-sweeping producer/consumer kernel (bf16 only)
void conv0_sweep_pack(const float* w, const float* scale, std::vector<float>& packed);
int launch_conv0_sweep(const Conv3dTileDesc& d, int dtype, hipStream_t s);      // dtype BF16 or F16

// conv0_sweep_x3.hip — the same for the BF16X3 mode: fp32 feature map in, split-pair c0 out, three MFMAs per product
int conv0_sweep_x3_upload(const std::vector<float>& packed, void** dev);      // packed: conv0_sweep_pack's fp32 fragment order
int launch_conv0_sweep_x3(const Conv3dTileDesc& d, hipStream_t s);           // d.feat: fp32 [V][H][W][32]; d.wgt: the uploaded array
int launch_bx3_to_f32(const void* in, float* out, long long n, hipStream_t s);   // misc_kernels.hip: split-pair tensor -> plain fp32 (n % 4 == 0)

// prob_sparse.hip — conv11 + skip + prob conv + softmax + depth on the neighbourhoods of the chosen pixels (bf16)
int launch_prob_sparse(const void* u9, const void* c0, const void* w11_packed, const float* bias11, const float* wprob,
                       const int* choose, const float* depths, float* prob, float* depth_out, int v0, int Vc, int B, int P,
                       int D, int H, int W, int dtype, hipStream_t s, const void* w11_taps = nullptr);
// conv11 weights [16][8][27] (x BN scale) -> the nine in-plane-tap A operands of prob_sparse2_kernel, fp32 [9][16][4][8] (prob_sparse.hip)
void prob_sparse_pack(const float* w, const float* scale, std::vector<float>& packed);

// prepare.hip — batched device-side AdaPoseEstimator_v5.prepare_model_input
int launch_prepare_inputs(const float* rgb, const unsigned char* mask, const double* K, const int* frame_map, int N, int H, int W, int S, int P,
                          unsigned seed, float* img, int* choose, float* pts2d, double* Kcrop, int* window, int* valid,
                          unsigned char* small_scratch, hipStream_t s, int frame0 = 0);

int launch_umeyama_ransac(const float* nocs, const float* depth, const int* choose, const double* Kc, const double* E1,
                          double* bbox, double* srt, int* valid, int B, int P, int img, unsigned seed, hipStream_t s);

int launch_projection(const double* Kc, const double* E, float* P, int n, hipStream_t s);      // prepare.hip
// pnp.hip — the use_depth: False tail of predict (NOCS matches -> triangulation -> scale -> EPnP-RANSAC -> VVS -> world bbox)
int launch_pnp_ransac(const float* nocs1, const float* pts1, const float* nocs2, const float* pts2, const double* K, const double* E1,
                      const double* E2, double* bbox, double* srt, int* info, int* valid, int B, int P, unsigned seed, hipStream_t s);

int launch_mask_extent(const unsigned char* mask, int N, int H, int W, int* ext, int* count, hipStream_t s);

// postproc.hip
int launch_postprocess(const float* nocs, const float* depth, const float* rot, const int* choose, const double* Kc,
                       const double* E1, double* bbox, double* ts_out, int* valid, int B, int P, int img, hipStream_t s,
                       void* scratch = nullptr, size_t scratch_bytes = 0);
// microbench.hip: achievable-peak probes for bench.py
int launch_microbench_mfma(float* scratch, int iters, int random_operands, double* flops, hipStream_t s);
int microbench_mfma_scratch_floats(int* n);
int launch_microbench_copy(const void* src, void* dst, size_t bytes, hipStream_t s);
size_t postprocess_scratch_bytes(int B);      // device scratch of the split (small-batch) form, per call
int postprocess_slices(int B);                // workgroups per pose the split form would use (1 = one-kernel form)

// ppo_kernels.hip
int launch_gae(int T, int N, const float* rewards, const unsigned char* dones, const float* values, const float* last_values,
               float gamma, float lam, float* returns, float* adv, double* sums, hipStream_t s);
int launch_adv_normalise(long long n_local, float* adv, const double* sums, double count_total, hipStream_t s);

// policy_kernels.hip — PPO actor-critic (flat fp32 parameter vector in the reference's state_dict order)
struct PolicyLayout { int dims[5]; int log_std; int w[2][4]; int b[2][4]; int total; };   // net 0 = actor, 1 = critic
struct PolicyOptState { int t; int n_updates; float lr; float last_kl; float last_norm; float pad_; double sum_surr; double sum_vloss; };
int launch_policy_forward(const float* params, const PolicyLayout& L, int n, int mode, const float* obs, const float* noise,
                          float* actions, float* logp, float* value, float* mu, hipStream_t s);
int policy_partial_floats(const PolicyLayout& L, int n);
int launch_ppo_minibatch(const float* params, const PolicyLayout& L, int n, const float* obs, const float* actions,
                         const float* old_logp, const float* adv, const float* returns, const float* old_values,
                         const float* old_mu, const float* old_sigma, float clip, float vcoef, float ecoef, float* partial,
                         float* grads, hipStream_t s);
int launch_ppo_adam(float* params, const float* grads, float* m, float* v, PolicyOptState* st, int total, float inv_world,
                    float max_norm, float desired_kl, float lr_min, float lr_max, int adaptive, hipStream_t s);

}  // namespace rgbm
