#pragma once
#include "kernels.h"
#include "layers.h"

namespace rgbm {

struct Arena {
  char* base; size_t off; size_t cap; size_t peak;
  explicit Arena(void* b, size_t c) : base((char*)b), off(0), cap(c), peak(0) {}
  void* alloc(size_t bytes) {
    off = align_up(off, 256);
    void* p = base ? base + off : nullptr;
    off += bytes;
    if (off > peak) peak = off;
    return p;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

struct AdaPose {
  int dtype = F32;
  // storage type of the pose MLP (pose_mlp1 / pose_mlp2, network_v5.py:470-494): fp16 for the bf16 throughput mode (its
  // activations are O(1); fp16 per-point features keep 11 mantissa bits, the means over points stay fp32, and the rotation
  // error stays at the 2e-3 the bf16 backbone sets), fp32 for fp32 nets and for the fp16 accuracy mode (where fp16 here would
  // raise the rotation error from 4e-5 to 3e-4)
  // storage type of the four per-point layers of the pose MLP: fp16 for bf16 nets, split pairs for split-pair nets (the exact-fp32
  // matrix path runs at a sixteenth of the bf16 rate: 1.5 ms per step for these layers), fp32 otherwise
  int pose_dtype() const { return dtype == BF16 ? F16 : (dtype == BF16X3 && pose_x3) ? BF16X3 : F32; }
  int pose_x3 = 1;
  int img = 224, n_pts = 1024, n_depth = 24;
  int img_cpad = 4;
  int max_chunk = 512;           // views per cost-volume chunk (bounds the workspace: ~80 MB per view in bf16; 512 = batch 256 in one chunk)

  struct Block { ConvLayer c1, c2, ds; bool has_ds = false; int stride = 1, planes = 0; };
  ConvLayer conv1;
  Block blocks[16];
  int n_blocks = 0;
  ConvLayer psp[4], up1, up2, up3, fin;
  UpConvLayer up1c, up2c;       // up_1 / up_2 as a low-resolution 1x1 GEMM + tap combination (upconv.hip)
  int sparse_dec = 2;           // with the sparse tail: 2 = every 3-D layer (and the plane sweep) only on the tiles inside the chosen pixels' dependency cones, 1 = conv7 / conv9 only, 0 = dense (A/B, and the c0 .. u9 taps)
  void* stem_w = nullptr;       // conv1 packed for the one-kernel stem (stem.hip; 16-bit and split-pair storage)
  int stem = 0;                 // 1: NCHW images -> conv1 7x7 + ReLU + max-pool in one kernel (default for 16-bit and split-pair storage, set in create()); 0: copy, implicit-GEMM conv, pool (materialises `conv1`)
  UpConvFinal tail;             // up_3 + final in one kernel (upconv_final.hip; 16-bit and split-pair storage)
  int upconv = 7;               // bit 0: up_1, bit 1: up_2 through UpConvLayer, bit 2: up_3 + final through UpConvFinal (16-bit / split pairs); 0 = x2 resize + 3x3 conv on the up-sampled grid, for A/B and tests
  ConvLayer c3d[7], dc[3];      // generic implicit-GEMM versions (kept for A/B: cost_impl = 0)
  // norm_mode 1 (per-sample BatchNorm3d: the as-shipped train-mode behaviour at batch 1): the same ten layers without
  // BN / activation, their gamma / beta, and the in-place normalisation of bn_kernels.hip behind each of them
  int norm_mode = 0;
  ConvLayer c3d_raw[7], dc_raw[3];
  float* bn_gamma[10] = {nullptr};
  float* bn_beta[10] = {nullptr};
  struct Tile3d { void* w = nullptr; float* bias = nullptr; int Cin = 0, Cout = 0; };
  Tile3d t3d[10];               // halo-tiled versions: 0..6 conv0..6, 7..9 conv7/9/11
  int igemm_conv6 = 1;          // bf16 + cost_impl 3: conv6 through the implicit-GEMM path instead of the halo-tile kernel
  int fuse_final = 1;           // bf16: PSPNet `final` 1x1 fused into up_3's kernel (0 = two launches; `u3` is then materialised)
  int view2_heads = 1;          // 0: the cost volume, the point heads and the pose regression run for the view-1 crops only (the view-2 outputs are
                                // filled with NaN): what `AdaPoseEstimator_v5.estimate` consumes — interface_v5.py:318-374 builds the box from
                                // view1_nocs / view1_depth / view1_r and drops the rest — at ~3/4 of the time; the backbone still runs on both views
  int sparse_tail = 1;          // cost_impl 3: evaluate conv11 + prob only where prob is gathered (0 = dense conv11, for A/B and tests)
  int sweep_f16 = 1;            // bf16 nets: `final` writes the feature map as f16 and the plane sweep blends it with packed f16 FMAs (conv0_sweep.hip; 0 = bf16 feature map, fp32 blend)
  void* w11_taps = nullptr;     // conv11 for the sparse tail's round-6 kernel: prob_sparse_pack's nine in-plane-tap operands (bf16x3: hi operands, then lo)
  void* sweep_w_f16 = nullptr;  // bf16 nets: the same conv0 weights as f16 (sweep_f16)
  void* sweep_w = nullptr;      // conv0 weights in conv0_sweep.hip fragment order (16-bit nets; bf16x3 nets: hi + lo operand arrays of conv0_sweep_x3.hip)
  int cost_impl = 3;            // 0 generic igemm + materialised volume, 1 tiled + materialised volume, 2 tiled + fused warp,
                                // 3 = 2 with the depth-sweeping conv0 kernel (bf16; fp32 nets run 2)
  void* w11_x3 = nullptr;       // bf16x3 nets: conv11 weights for prob_sparse (16-bit step geometry, hi operands then lo operands)
  float* wprob = nullptr;
  ConvLayer inst, nh[3], npm[2], pm1[2], pm2[2];
  float* pm2_0_wfull = nullptr;
  float* pmlp_table = nullptr;    // the per-point NOCS branch's six layers as point_mlp_kernel's LDS image (kernels.h: point_mlp_pack)
  float* pm2_0_bias = nullptr;
  float* head_w[3][3] = {{nullptr}};
  float* head_b[3][3] = {{nullptr}};

  struct Buffers {
    float *Pviews, *homog; int* choose; void* feat;
    unsigned char* masks;           // sparse cost regularisation: tile masks of the 3-D layers per view of a cost-volume chunk (prob_sparse.hip)
    int *sweep_list, *sweep_count;  // ... and the needed tiles of the depth-sweeping conv0
    float* featf;                      // bf16x3 nets: plain fp32 copy of feat (what the plane sweep and the point heads gather from)
    float *X0, *X1, *H128, *H64, *nocs4, *N32, *PF96, *prob, *depth, *Q128a, *Q128b, *G256a, *G256b;
    void* PF96h;                       // fp16 copy of PF96 (pose MLP input of 16-bit nets)
    float *glob, *vbias, *pf2, *h1, *h2, *r6, *R, *tv, *sv;
    void *imgpad, *c1, *lb[4], *pooled[4], *stage[4], *cat, *ups, *u1, *u2, *u3;
    void *vol, *c[7], *u7, *u9, *u11;
    void* bn_scratch;
  };
  struct Outputs {   // device fp32, reference shapes (network_v5.py:510-515)
    float *nocs1, *nocs2;   // [B,P,3]
    float *depth1, *depth2; // [B,P]
    float *r1, *r2;         // [B,3,3]
    float *t1, *t2, *s1, *s2;  // [B,3]
  };

  int create(const StateDict& sd, int dtype, int norm_mode = 0);
  void destroy();
  size_t workspace_bytes(int B) const;
  int forward(int B, const float* img1, const float* img2, const int* choose1, const int* choose2, const float* P1,
              const float* P2, const float* depths, void* workspace, size_t workspace_size, const Outputs& out,
              hipStream_t s, int stop_after = 0) const;
  mutable int last_f_index = 0;   // which rotating buffer holds the layer4 output (debug fetch)

  // exposed for layer-level tests
  int plan(int B, Arena& A, Buffers& bf) const;
  int pspnet(const Buffers& bf, int V, const float* img1, const float* img2, hipStream_t s) const;
  int cost_volume(const Buffers& bf, int V, int B, const float* depths, hipStream_t s) const;
  int chunk_views(int V) const;
  // bf16x3 on the default path: nothing reads the split-pair feature map (the sweep and the point heads gather from plain fp32), so
  // `final` writes fp32 directly; the A/B paths (materialised volume, halo-tile conv0, per-sample BN) still want split pairs
  bool feat_f32_only() const;
  bool feat_f16() const;        // bf16 nets: the feature map (bf.feat) holds f16 in this forward (option sweep_f16 and the paths that can produce / consume it)
  // the conditions under which cost_volume() skips tiles outside the chosen pixels' dependency cones (option sparse_dec)
  bool sparse_active() const;
};

const char* last_error_cstr();

}  // namespace rgbm
