// AdaPose stereo pose network forward, orchestrated on one HIP stream.
// Follows /root/reference/models/pose_estimator/AdaPose/lib/network_v5.py:418-519 (eval mode) with an
// MI355X-first dataflow: channels-last tensors, BN folded into the 3-D convs, the probability conv and the
// depth-guided fusion evaluated only at the 1024 sampled pixels, the cost volume processed in view chunks
// so its footprint is bounded, the second half of pose_mlp2's first layer (global feature) turned into a
// per-view bias.  Views are ordered v = side*B + b (all view-1 images, then all view-2 images).
#include "adapose.h"

#include <math.h>
#include <string.h>

namespace rgbm {

static const int kLayerPlanes[4] = {64, 128, 256, 512};
static const int kLayerBlocks[4] = {3, 4, 6, 3};
static const int kLayerStride[4] = {1, 2, 1, 1};
static const int kLayerDil[4] = {1, 1, 2, 4};
static const int kPspBins[4] = {1, 2, 3, 6};
static const int kPspSmallViews = 32;      // up to this many views the PSP stage pools and multiplies its 50 cells per view in one launch (see AdaPose::pspnet)


static const HostTensor* find(const StateDict& sd, const std::string& name) {
  auto it = sd.find(name);
  return it == sd.end() ? nullptr : &it->second;
}

#define GET(var, name)                                               \
  const HostTensor* var = find(sd, name);                            \
  RGBM_REQUIRE(var != nullptr, std::string("missing weight ") + (name))

static int init_conv2d(ConvLayer& L, int dtype, const StateDict& sd, const std::string& wname, const char* bname, int Cin,
                       int Cout, int k, int stride, int pad, int dil, int act, float slope, int Cin_pad) {
  GET(w, wname);
  RGBM_REQUIRE(w->numel() == (long long)Cout * Cin * k * k, "weight shape " + wname);
  const float* b = nullptr;
  if (bname) { GET(bt, std::string(bname)); b = bt->data; }
  ConvGeom g;
  g.Cin = Cin; g.Cout = Cout; g.KH = g.KW = k; g.sh = g.sw = stride; g.ph = g.pw = pad; g.dilh = g.dilw = dil;
  g.act = act; g.slope = slope;
  return L.init(dtype, g, w->data, b, nullptr, nullptr, Cin_pad, Cout);
}

static int bn_fold(const StateDict& sd, const std::string& p, int C, std::vector<float>& scale, std::vector<float>& shift) {
  GET(gm, p + "weight"); GET(bt, p + "bias"); GET(mu, p + "running_mean"); GET(var, p + "running_var");
  RGBM_REQUIRE(gm->numel() == C && bt->numel() == C && mu->numel() == C && var->numel() == C, "bn shape " + p);
  scale.resize(C); shift.resize(C);
  for (int i = 0; i < C; ++i) {
    const float inv = 1.0f / sqrtf(var->data[i] + 1e-5f);
    scale[i] = gm->data[i] * inv;
    shift[i] = bt->data[i] - mu->data[i] * scale[i];
  }
  return 0;
}

static int init_conv3d_bn(ConvLayer& L, int dtype, const StateDict& sd, const std::string& p, int Cin, int Cout, int stride,
                          bool transposed) {
  GET(w, p + "conv.weight");
  RGBM_REQUIRE(w->numel() == (long long)Cout * Cin * 27, "weight shape " + p);
  std::vector<float> scale, shift;
  if (int rc = bn_fold(sd, p + "bn.", Cout, scale, shift)) return rc;
  ConvGeom g;
  g.Cin = Cin; g.Cout = Cout; g.KD = g.KH = g.KW = 3; g.sd = g.sh = g.sw = stride; g.pd = g.ph = g.pw = 1;
  g.transposed = transposed; g.act = ACT_RELU;
  return L.init(dtype, g, w->data, nullptr, scale.data(), shift.data(), Cin, Cout);
}

static int init_linear(ConvLayer& L, const StateDict& sd, const std::string& p, int Cin, int Cout, int act, int Cin_pad,
                       int Cout_pad, const float* w_override = nullptr, int ldtype = F32) {
  GET(w, p + ".weight"); GET(b, p + ".bias");
  ConvGeom g;
  g.Cin = Cin; g.Cout = Cout; g.act = act;
  return L.init(ldtype, g, w_override ? w_override : w->data, b->data, nullptr, nullptr, Cin_pad, Cout_pad);
}

int AdaPose::create(const StateDict& sd, int dtype_, int norm_mode_) {
  dtype = dtype_;
  norm_mode = norm_mode_;
  const int E = dtype_chunk(dtype);
  img_cpad = E;                              // RGB padded to one 16-byte chunk
  const std::string fe = "img_extractor.feats.";
  if (int rc = init_conv2d(conv1, dtype, sd, fe + "conv1.weight", nullptr, 3, 64, 7, 2, 3, 1, ACT_RELU, 0.f, img_cpad)) return rc;
  if (dtype != F32) {
    GET(w1, fe + "conv1.weight");
    RGBM_REQUIRE(w1->numel() == 64 * 3 * 7 * 7, "conv1 shape");
    std::vector<float> pk;
    stem_pack(w1->data, pk);
    if (int rc = upload_packed(pk, dtype, &stem_w)) return rc;
    // same-box A/B at batch 256 (tools/ab_option.py <dtype> stem 0 1): bf16 forward 56.58 -> 55.62 ms; split pairs 125.06 -> 124.68 ms
    // with 4-row tiles (8-row tiles: 79 KB of staging, 122.63 -> 122.91 ms)
    stem = 1;
  }
  int inpl = 64, nb = 0;
  for (int li = 0; li < 4; ++li) {
    for (int b = 0; b < kLayerBlocks[li]; ++b) {
      Block& blk = blocks[nb++];
      const int planes = kLayerPlanes[li];
      const int cin = b == 0 ? inpl : planes;
      const int stride = b == 0 ? kLayerStride[li] : 1;
      const int dil = b == 0 ? 1 : kLayerDil[li];      // block 0 never dilated (pspnet.py:59-62)
      const std::string p = fe + "layer" + std::to_string(li + 1) + "." + std::to_string(b) + ".";
      if (int rc = init_conv2d(blk.c1, dtype, sd, p + "conv1.weight", nullptr, cin, planes, 3, stride, dil, dil, ACT_RELU, 0.f, cin)) return rc;
      if (int rc = init_conv2d(blk.c2, dtype, sd, p + "conv2.weight", nullptr, planes, planes, 3, 1, dil, dil, ACT_RELU, 0.f, planes)) return rc;
      blk.has_ds = find(sd, p + "downsample.0.weight") != nullptr;
      if (blk.has_ds)
        if (int rc = init_conv2d(blk.ds, dtype, sd, p + "downsample.0.weight", nullptr, cin, planes, 1, stride, 0, 1, ACT_NONE, 0.f, cin)) return rc;
      blk.stride = stride; blk.planes = planes;
    }
    inpl = kLayerPlanes[li];
  }
  n_blocks = nb;
  for (int i = 0; i < 4; ++i)
    if (int rc = init_conv2d(psp[i], dtype, sd, "img_extractor.psp.stages." + std::to_string(i) + ".1.weight", nullptr, 512, 128, 1, 1, 0, 1, ACT_RELU, 0.f, 512)) return rc;
  struct { ConvLayer* L; const char* nm; int cin, cout; } ups[3] = {{&up1, "up_1", 1024, 256}, {&up2, "up_2", 256, 64}, {&up3, "up_3", 64, 64}};
  for (auto& u : ups) {
    const std::string p = std::string("img_extractor.") + u.nm + ".conv.";
    GET(sl, p + "1.weight");
    const std::string bn = p + "0.bias";
    if (int rc = init_conv2d(*u.L, dtype, sd, p + "0.weight", bn.c_str(), u.cin, u.cout, 3, 1, 1, 1, ACT_PRELU, sl->data[0], u.cin)) return rc;
    UpConvLayer* uc = u.L == &up1 ? &up1c : u.L == &up2 ? &up2c : nullptr;
    if (uc) {
      GET(w, p + "0.weight"); GET(b, bn);
      if (int rc = uc->init(dtype, u.cin, u.cout, w->data, b->data, ACT_PRELU, sl->data[0])) return rc;
    }
  }
  if (int rc = init_conv2d(fin, dtype, sd, "img_extractor.final.weight", "img_extractor.final.bias", 64, 32, 1, 1, 0, 1, ACT_NONE, 0.f, 64)) return rc;
  if (dtype != F32) {
    GET(w3, "img_extractor.up_3.conv.0.weight"); GET(b3, "img_extractor.up_3.conv.0.bias"); GET(s3, "img_extractor.up_3.conv.1.weight");
    GET(wf, "img_extractor.final.weight"); GET(bfin, "img_extractor.final.bias");
    if (int rc = tail.init(dtype, w3->data, b3->data, s3->data[0], wf->data, bfin->data)) return rc;
  }


  const std::string cr = "cost_regularization.";
  const int crc[8] = {32, 8, 16, 16, 32, 32, 64, 64};
  const int crs[7] = {1, 2, 1, 2, 1, 2, 1};
  for (int i = 0; i < 7; ++i)
    if (int rc = init_conv3d_bn(c3d[i], dtype, sd, cr + "conv" + std::to_string(i) + ".", crc[i], crc[i + 1], crs[i], false)) return rc;
  if (int rc = init_conv3d_bn(dc[0], dtype, sd, cr + "conv7.", 64, 32, 2, true)) return rc;
  if (int rc = init_conv3d_bn(dc[1], dtype, sd, cr + "conv9.", 32, 16, 2, true)) return rc;
  if (int rc = init_conv3d_bn(dc[2], dtype, sd, cr + "conv11.", 16, 8, 2, true)) return rc;
  if (norm_mode == 1) {
    const char* nm[10] = {"conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11"};
    for (int i = 0; i < 10; ++i) {
      const bool tr = i >= 7;
      const int cin = tr ? (i == 7 ? 64 : i == 8 ? 32 : 16) : crc[i], cout = tr ? cin / 2 : crc[i + 1];
      const std::string p = cr + nm[i] + ".";
      GET(w, p + "conv.weight"); GET(gm, p + "bn.weight"); GET(bt, p + "bn.bias");
      RGBM_REQUIRE(w->numel() == (long long)cout * cin * 27 && gm->numel() == cout && bt->numel() == cout, "weight shape " + p);
      ConvGeom g;
      g.Cin = cin; g.Cout = cout; g.KD = g.KH = g.KW = 3; g.sd = g.sh = g.sw = tr ? 2 : crs[i]; g.pd = g.ph = g.pw = 1;
      g.transposed = tr; g.act = ACT_NONE;
      ConvLayer& L = tr ? dc_raw[i - 7] : c3d_raw[i];
      if (int rc = L.init(dtype, g, w->data, nullptr, nullptr, nullptr, cin, cout)) return rc;
      if (upload_f32(gm->data, cout, &bn_gamma[i])) return -2;
      if (upload_f32(bt->data, cout, &bn_beta[i])) return -2;
    }
  }
  {
    // halo-tiled versions of the same ten layers (conv3d_tile.hip)
    const char* nm[10] = {"conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11"};
    const int cin[10] = {32, 8, 16, 16, 32, 32, 64, 64, 32, 16};
    const int cout[10] = {8, 16, 16, 32, 32, 64, 64, 32, 16, 8};
    for (int i = 0; i < 10; ++i) {
      const std::string p = cr + nm[i] + ".";
      GET(w, p + "conv.weight");
      std::vector<float> scale, shift, packed;
      if (int rc = bn_fold(sd, p + "bn.", cout[i], scale, shift)) return rc;
      const int coutp = cout[i] < 16 ? 16 : cout[i];
      conv3d_tile_pack(w->data, scale.data(), cin[i], cout[i], coutp, i >= 7, dtype, packed);
      t3d[i].Cin = cin[i]; t3d[i].Cout = cout[i];
      if (upload_packed(packed, dtype, &t3d[i].w)) return -2;
      std::vector<float> bpad(coutp, 0.f);
      for (int o = 0; o < cout[i]; ++o) bpad[o] = shift[o];
      if (upload_f32(bpad.data(), bpad.size(), &t3d[i].bias)) return -2;
      if (i == 9 && dtype != F32) {
        // conv11 for the sparse tail (prob_sparse2_kernel): one operand per in-plane tap, both depth parities on the 16 MFMA rows
        std::vector<float> pt;
        prob_sparse_pack(w->data, scale.data(), pt);
        if (dtype == BF16X3) { if (conv0_sweep_x3_upload(pt, &w11_taps)) return -2; }
        else if (upload_packed(pt, dtype, &w11_taps)) return -2;
      }
      if (i == 9 && dtype == BF16X3) {
        // conv11 for the sparse tail (prob_sparse.hip): its step structure is the 16-bit kernels' (two taps x 16 channels per
        // MFMA), so the weights are packed in that geometry and split into hi / lo operand arrays
        std::vector<float> p16;
        conv3d_tile_pack(w->data, scale.data(), cin[i], cout[i], coutp, true, BF16, p16);
        if (conv0_sweep_x3_upload(p16, &w11_x3)) return -2;
      }
      if (i == 0 && dtype != F32) {
        // the same conv0 weights in the depth-sweeping kernel's paired-tap fragment order (conv0_sweep.hip / conv0_sweep_x3.hip)
        conv0_sweep_pack(w->data, scale.data(), packed);
        if (dtype == BF16X3) { if (conv0_sweep_x3_upload(packed, &sweep_w)) return -2; }
        else if (upload_packed(packed, dtype, &sweep_w)) return -2;
        // the f16 form of the sweep (sweep_f16) only for weights that f16 can hold: otherwise sweep_w_f16 stays null and feat_f16() is false
        if (dtype == BF16 && weights_fit_f16(packed) && upload_packed(packed, F16, &sweep_w_f16)) return -2;
      }
    }
  }
  {
    GET(w, cr + "prob.weight");
    RGBM_REQUIRE(w->numel() == 8 * 27, "prob weight shape");
    std::vector<float> wp(27 * 8);
    for (int c = 0; c < 8; ++c) for (int t = 0; t < 27; ++t) wp[t * 8 + c] = w->data[c * 27 + t];
    if (upload_f32(wp.data(), wp.size(), &wprob)) return -2;
  }
  // fp32 point heads as 1x1 convs
  if (int rc = init_linear(inst, sd, "instance_color.0", 32, 64, ACT_RELU, 32, 64)) return rc;
  if (int rc = init_linear(nh[0], sd, "nocs_head.0", 64, 128, ACT_RELU, 64, 128)) return rc;
  if (int rc = init_linear(nh[1], sd, "nocs_head.2", 128, 64, ACT_RELU, 128, 64)) return rc;
  if (int rc = init_linear(nh[2], sd, "nocs_head.4", 64, 3, ACT_TANH, 64, 4)) return rc;
  if (int rc = init_linear(npm[0], sd, "nocs_pts_mlp.0", 3, 32, ACT_RELU, 4, 32)) return rc;
  if (int rc = init_linear(npm[1], sd, "nocs_pts_mlp.2", 32, 64, ACT_RELU, 32, 64)) return rc;
  {
    // the same six layers as one LDS image for point_mlp_kernel (head_kernels.hip)
    static const char* const names[6] = {"instance_color.0", "nocs_head.0", "nocs_head.2", "nocs_head.4", "nocs_pts_mlp.0", "nocs_pts_mlp.2"};
    static const int cin[6] = {32, 64, 128, 64, 3, 32}, cout[6] = {64, 128, 64, 3, 32, 64};
    const float* w[6]; const float* b[6];
    for (int l = 0; l < 6; ++l) {
      GET(wt, std::string(names[l]) + ".weight"); GET(bt, std::string(names[l]) + ".bias");
      RGBM_REQUIRE(wt->numel() == (long long)cin[l] * cout[l] && bt->numel() == cout[l], std::string("point MLP shape ") + names[l]);
      w[l] = wt->data; b[l] = bt->data;
    }
    std::vector<float> table((size_t)point_mlp_table_floats());
    point_mlp_pack(w, b, table.data());
    if (upload_f32(table.data(), table.size(), &pmlp_table)) return -2;
  }
  if (int rc = init_linear(pm1[0], sd, "pose_mlp1.0", 96, 128, ACT_RELU, 96, 128, nullptr, pose_dtype())) return rc;
  if (int rc = init_linear(pm1[1], sd, "pose_mlp1.2", 128, 128, ACT_RELU, 128, 128, nullptr, pose_dtype())) return rc;
  {
    // pose_mlp2.0 sees cat(point feature[128], global mean[128]); the global half becomes a per-view bias
    GET(w, "pose_mlp2.0.weight"); GET(b, "pose_mlp2.0.bias");
    RGBM_REQUIRE(w->numel() == 256 * 256, "pose_mlp2.0 shape");
    std::vector<float> wl(256 * 128);
    for (int o = 0; o < 256; ++o) for (int i = 0; i < 128; ++i) wl[o * 128 + i] = w->data[o * 256 + i];
    if (int rc = init_linear(pm2[0], sd, "pose_mlp2.0", 128, 256, ACT_RELU, 128, 256, wl.data(), pose_dtype())) return rc;
    if (upload_f32(w->data, 256 * 256, &pm2_0_wfull)) return -2;
    if (upload_f32(b->data, 256, &pm2_0_bias)) return -2;
  }
  if (int rc = init_linear(pm2[1], sd, "pose_mlp2.2", 256, 256, ACT_RELU, 256, 256, nullptr, pose_dtype())) return rc;
  const char* hn[3] = {"rotation_estimator", "translation_estimator", "size_estimator"};
  const int hout[3] = {6, 3, 3};
  for (int h = 0; h < 3; ++h) {
    const int dims[4] = {256, 256, 128, hout[h]};
    for (int l = 0; l < 3; ++l) {
      GET(w, std::string(hn[h]) + "." + std::to_string(2 * l) + ".weight");
      GET(b, std::string(hn[h]) + "." + std::to_string(2 * l) + ".bias");
      RGBM_REQUIRE(w->numel() == (long long)dims[l] * dims[l + 1], "head weight shape");
      if (upload_f32(w->data, w->numel(), &head_w[h][l])) return -2;
      if (upload_f32(b->data, b->numel(), &head_b[h][l])) return -2;
    }
  }
  return 0;
}

void AdaPose::destroy() {
  conv1.destroy();
  if (stem_w) (void)hipFree(stem_w);
  stem_w = nullptr;
  for (int i = 0; i < n_blocks; ++i) { blocks[i].c1.destroy(); blocks[i].c2.destroy(); if (blocks[i].has_ds) blocks[i].ds.destroy(); }
  for (auto& l : psp) l.destroy();
  up1.destroy(); up2.destroy(); up3.destroy(); fin.destroy();
  up1c.destroy(); up2c.destroy(); tail.destroy();
  for (auto& l : c3d) l.destroy();
  for (auto& l : dc) l.destroy();
  for (auto& l : c3d_raw) l.destroy();
  for (auto& l : dc_raw) l.destroy();
  for (int i = 0; i < 10; ++i) { if (bn_gamma[i]) (void)hipFree(bn_gamma[i]); if (bn_beta[i]) (void)hipFree(bn_beta[i]); bn_gamma[i] = bn_beta[i] = nullptr; }
  inst.destroy();
  for (auto& l : nh) l.destroy();
  for (auto& l : npm) l.destroy();
  for (auto& l : pm1) l.destroy();
  for (auto& l : pm2) l.destroy();
  for (auto& t : t3d) { if (t.w) (void)hipFree(t.w); if (t.bias) (void)hipFree(t.bias); t.w = nullptr; t.bias = nullptr; }
  if (sweep_w) (void)hipFree(sweep_w);
  sweep_w = nullptr;
  if (sweep_w_f16) (void)hipFree(sweep_w_f16);
  sweep_w_f16 = nullptr;
  if (w11_x3) (void)hipFree(w11_x3);
  w11_x3 = nullptr;
  if (w11_taps) (void)hipFree(w11_taps);
  w11_taps = nullptr;
  if (wprob) (void)hipFree(wprob);
  if (pm2_0_wfull) (void)hipFree(pm2_0_wfull);
  if (pmlp_table) { (void)hipFree(pmlp_table); pmlp_table = nullptr; }
  if (pm2_0_bias) (void)hipFree(pm2_0_bias);
  for (int h = 0; h < 3; ++h) for (int l = 0; l < 3; ++l) { if (head_w[h][l]) (void)hipFree(head_w[h][l]); if (head_b[h][l]) (void)hipFree(head_b[h][l]); }
}

bool AdaPose::feat_f32_only() const {
  return dtype == BF16X3 && cost_impl == 3 && sweep_w != nullptr && norm_mode == 0 && !(g_debug_flags & 4096);
}

// bf16 nets: what `final` writes and what the plane sweep and the point heads read is f16 - same bytes, three more mantissa bits, and the
// sweep's blend becomes four v_pk_fma_f16 per dword.  Needs the one-kernel up_3 + final (its epilogue knows the f16 form) and the
// depth-sweeping conv0 (the halo-tile / volume paths read the storage type).
bool AdaPose::feat_f16() const {
  return dtype == BF16 && sweep_f16 != 0 && cost_impl == 3 && sweep_w_f16 != nullptr && norm_mode == 0 && !(g_debug_flags & 4096) && (upconv & 4) &&
         tail.ready() && tail.f16_ready();
}

bool AdaPose::sparse_active() const {
  const bool b16 = dtype_size(dtype) == 2;
  return sparse_dec != 0 && norm_mode == 0 && cost_impl == 3 && (b16 || (dtype == BF16X3 && w11_x3)) && sparse_tail && sweep_w != nullptr &&
         !(g_debug_flags & 4096);
}

int AdaPose::chunk_views(int V) const {
  const int cap = norm_mode == 1 && max_chunk > 32 ? 32 : max_chunk;      // per-sample BN materialises the 32-channel volume (154 MB per view in fp32)
  return V < cap ? V : cap;
}

// Shared by workspace_bytes() (base == nullptr) and forward(): identical allocation order => identical offsets.
int AdaPose::plan(int B, Arena& A, Buffers& bf) const {
  const int V = 2 * B, P = n_pts, S = img;
  const size_t es = dtype_size(dtype);
  const size_t VP = (size_t)V * P;
  bf.Pviews = (float*)A.alloc((size_t)V * 16 * 4);
  bf.homog = (float*)A.alloc((size_t)V * 12 * 4);
  bf.choose = (int*)A.alloc(VP * 4);
  bf.masks = (unsigned char*)A.alloc((size_t)V * sparse_mask_bytes_per_view(img));
  bf.sweep_list = (int*)A.alloc((size_t)V * ((img + 11) / 12) * ((img + 15) / 16) * 4);
  bf.sweep_count = (int*)A.alloc((size_t)(V + 64) * 4);      // total, then one count per view
  bf.feat = A.alloc((size_t)V * S * S * 32 * es);
  bf.featf = dtype == BF16X3 ? (float*)A.alloc((size_t)V * S * S * 32 * 4) : nullptr;
  bf.X0 = (float*)A.alloc(VP * 32 * 4);
  bf.X1 = (float*)A.alloc(VP * 64 * 4);
  bf.H128 = (float*)A.alloc(VP * 128 * 4);
  bf.H64 = (float*)A.alloc(VP * 64 * 4);
  bf.nocs4 = (float*)A.alloc(VP * 4 * 4);
  bf.N32 = (float*)A.alloc(VP * 32 * 4);
  bf.PF96 = (float*)A.alloc(VP * 96 * 4);
  bf.PF96h = pose_dtype() == F16 ? A.alloc(VP * 96 * 2) : nullptr;
  bf.prob = (float*)A.alloc(VP * n_depth * 4);
  bf.depth = (float*)A.alloc(VP * 4);
  bf.Q128a = (float*)A.alloc(VP * 128 * 4);
  bf.Q128b = (float*)A.alloc(VP * 128 * 4);
  bf.G256a = (float*)A.alloc(VP * 256 * 4);
  bf.G256b = (float*)A.alloc(VP * 256 * 4);
  bf.glob = (float*)A.alloc((size_t)V * 128 * 4);
  bf.vbias = (float*)A.alloc((size_t)V * 256 * 4);
  bf.pf2 = (float*)A.alloc((size_t)V * 256 * 4);
  bf.h1 = (float*)A.alloc((size_t)V * 256 * 4);
  bf.h2 = (float*)A.alloc((size_t)V * 128 * 4);
  bf.r6 = (float*)A.alloc((size_t)V * 8 * 4);
  bf.R = (float*)A.alloc((size_t)V * 9 * 4);
  bf.tv = (float*)A.alloc((size_t)V * 4 * 4);
  bf.sv = (float*)A.alloc((size_t)V * 4 * 4);
  const size_t m0 = A.mark();
  // ---- phase A: PSPNet ----
  bf.imgpad = A.alloc((size_t)V * S * S * img_cpad * es);
  bf.c1 = A.alloc((size_t)V * (S / 2) * (S / 2) * 64 * es);
  const size_t lbuf = (size_t)V * ((size_t)(S / 4) * (S / 4) * 64 > (size_t)(S / 8) * (S / 8) * 512 ? (size_t)(S / 4) * (S / 4) * 64 : (size_t)(S / 8) * (S / 8) * 512) * es;
  for (int i = 0; i < 4; ++i) bf.lb[i] = A.alloc(lbuf);
  for (int i = 0; i < 4; ++i) {
    bf.pooled[i] = A.alloc((size_t)V * kPspBins[i] * kPspBins[i] * 512 * es);
    bf.stage[i] = A.alloc((size_t)V * kPspBins[i] * kPspBins[i] * 128 * es);
  }
  const size_t f8 = (size_t)(S / 8) * (S / 8), f4 = (size_t)(S / 4) * (S / 4), f2 = (size_t)(S / 2) * (S / 2), f1 = (size_t)S * S;
  bf.cat = A.alloc((size_t)V * f8 * 1024 * es);
  bf.ups = A.alloc((size_t)V * f4 * 1024 * es);      // == f2*256 == f1*64
  bf.u1 = A.alloc((size_t)V * f4 * 256 * es);
  bf.u2 = A.alloc((size_t)V * f2 * 64 * es);
  bf.u3 = A.alloc((size_t)V * f1 * 64 * es);
  A.release(m0);
  // ---- phase B: cost volume, per chunk of views ----
  const int Vc = chunk_views(V);
  const int D = n_depth;
  const size_t vox = (size_t)D * S * S;
  bf.vol = (cost_impl >= 2 && norm_mode == 0) ? nullptr : A.alloc((size_t)Vc * vox * 32 * es);   // fused-warp conv0 never materialises it
  bf.bn_scratch = norm_mode == 1 ? A.alloc(bn_scratch_bytes(Vc)) : nullptr;
  bf.c[0] = A.alloc((size_t)Vc * vox * 8 * es);
  bf.c[1] = A.alloc((size_t)Vc * (vox / 8) * 16 * es);
  bf.c[2] = A.alloc((size_t)Vc * (vox / 8) * 16 * es);
  bf.c[3] = A.alloc((size_t)Vc * (vox / 64) * 32 * es);
  bf.c[4] = A.alloc((size_t)Vc * (vox / 64) * 32 * es);
  bf.c[5] = A.alloc((size_t)Vc * (vox / 512) * 64 * es);
  bf.c[6] = A.alloc((size_t)Vc * (vox / 512) * 64 * es);
  bf.u7 = A.alloc((size_t)Vc * (vox / 64) * 32 * es);
  bf.u9 = A.alloc((size_t)Vc * (vox / 8) * 16 * es);
  bf.u11 = A.alloc((size_t)Vc * vox * 8 * es);
  A.release(m0);
  return 0;
}

size_t AdaPose::workspace_bytes(int B) const {
  Arena A(nullptr, 0);
  Buffers bf;
  plan(B, A, bf);
  return A.peak + 256;
}

int AdaPose::pspnet(const Buffers& bf, int V, const float* img1, const float* img2, hipStream_t s) const {
  const int S = img;
  int H = S / 2, W = S / 2;
  if (stem && stem_w) {
    // NCHW fp32 images -> conv1 + ReLU + max-pool in one kernel (no padded copy, no 112 x 112 x 64 tensor)
    if (int rc = launch_stem(dtype, img1, img2, stem_w, bf.lb[0], V / 2, V, S, s)) return rc;
  } else {
    const size_t es = dtype_size(dtype);
    const int B = V / 2;
    if (int rc = launch_nchw_to_nhwc_pad(dtype, img1, bf.imgpad, B, 3, S, S, img_cpad, s)) return rc;
    if (int rc = launch_nchw_to_nhwc_pad(dtype, img2, (char*)bf.imgpad + (size_t)B * S * S * img_cpad * es, B, 3, S, S, img_cpad, s)) return rc;
    if (int rc = conv1.run(bf.imgpad, bf.c1, V, 1, S, S, 64, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = launch_maxpool3x3s2(dtype, bf.c1, bf.lb[0], V, H, W, 64, s)) return rc;
  }
  H = S / 4; W = S / 4;
  int xi = 0;                                   // index of the buffer holding x
  for (int i = 0; i < n_blocks; ++i) {
    const Block& blk = blocks[i];
    void* x = bf.lb[xi];
    void* t = bf.lb[(xi + 1) & 3];
    void* r = bf.lb[(xi + 2) & 3];
    void* y = bf.lb[(xi + 3) & 3];
    int Do, Ho, Wo;
    blk.c1.out_dims(1, H, W, Do, Ho, Wo);
    if (int rc = blk.c1.run(x, t, V, 1, H, W, blk.planes, nullptr, 0, nullptr, 0, s)) return rc;
    const void* res = x;
    if (blk.has_ds) {
      if (int rc = blk.ds.run(x, r, V, 1, H, W, blk.planes, nullptr, 0, nullptr, 0, s)) return rc;
      res = r;
    }
    if (int rc = blk.c2.run(t, y, V, 1, Ho, Wo, blk.planes, res, RES_PRE_ACT, nullptr, 0, s)) return rc;
    H = Ho; W = Wo;
    xi = (xi + 3) & 3;
  }
  const void* f = bf.lb[xi];                    // [V][H][W][512], H = W = S/8
  last_f_index = xi;
  RGBM_REQUIRE(H == S / 8 && W == S / 8, "feature stride");
  if (g_debug_flags & 1024) {      // the PSP stage as rounds 1-5 ran it: copy, pooling, four GEMM launches, four resizes
    if (int rc = launch_copy_channels(dtype, f, bf.cat, (long long)V * H * W, 512, 1024, 0, s)) return rc;
    {
      void* outs[4] = {bf.pooled[0], bf.pooled[1], bf.pooled[2], bf.pooled[3]};
      if (int rc = launch_adaptive_avgpool_multi(dtype, f, outs, kPspBins, 4, V, H, W, 512, s)) return rc;     // all four bin sizes, one launch
    }
    for (int i = 0; i < 4; ++i) {
      const int Sb = kPspBins[i];
      if (int rc = psp[i].run(bf.pooled[i], bf.stage[i], V, 1, Sb, Sb, 128, nullptr, 0, nullptr, 0, s)) return rc;
      if (int rc = launch_resize_bilinear_ac(dtype, bf.stage[i], bf.cat, V, Sb, Sb, 128, H, W, 1024, 512 + 128 * i, s)) return rc;
    }
  } else {
    // Small batches (the deployment shape, B = 1 .. 8): pooling and the four 512 -> 128 convs of the 50 cells per view in one launch on the
    // vector pipe (misc_kernels.hip; the GEMM launches it replaces had 2 .. 576 rows); larger ones keep the pooling launch and the
    // implicit GEMMs.  The concat (backbone channels + the four resized stages) is one launch at any size.
    bool small = V <= kPspSmallViews;
    for (int i = 0; i < 4; ++i)
      small = small && psp[i].packs.size() == 1 && psp[i].packs[0].Kpad == 512 && psp[i].Cout_pad == 128 && psp[i].bias == nullptr;
    if (small) {
      const void* w[4] = {psp[0].packs[0].w, psp[1].packs[0].w, psp[2].packs[0].w, psp[3].packs[0].w};
      void* outs[4] = {bf.stage[0], bf.stage[1], bf.stage[2], bf.stage[3]};
      if (int rc = launch_psp_pool_conv(dtype, f, 512, w, outs, kPspBins, V, H, W, psp[0].g.act, psp[0].g.slope, s)) return rc;
    } else {
      void* outs[4] = {bf.pooled[0], bf.pooled[1], bf.pooled[2], bf.pooled[3]};
      if (int rc = launch_adaptive_avgpool_multi(dtype, f, outs, kPspBins, 4, V, H, W, 512, s)) return rc;     // all four bin sizes, one launch
      for (int i = 0; i < 4; ++i)
        if (int rc = psp[i].run(bf.pooled[i], bf.stage[i], V, 1, kPspBins[i], kPspBins[i], 128, nullptr, 0, nullptr, 0, s)) return rc;
    }
    const void* stages[4] = {bf.stage[0], bf.stage[1], bf.stage[2], bf.stage[3]};
    if (int rc = launch_psp_resize_cat(dtype, f, stages, kPspBins, bf.cat, V, H, W, s)) return rc;
  }
  // up_1 / up_2: 1x1 GEMM at the low resolution (nine taps stacked on the output channels, z in `ups`: 9/16 of the up-sampled
  // tensor it replaces) + tap combination; or, for A/B, the x2 resize followed by the 3x3 conv on the up-sampled grid
  if (upconv & 1) {
    if (int rc = up1c.run(bf.cat, bf.ups, bf.u1, V, H, W, 256, s)) return rc;
  } else {
    if (int rc = launch_resize_bilinear_ac(dtype, bf.cat, bf.ups, V, H, W, 1024, 2 * H, 2 * W, 1024, 0, s)) return rc;
    if (int rc = up1.run(bf.ups, bf.u1, V, 1, 2 * H, 2 * W, 256, nullptr, 0, nullptr, 0, s)) return rc;
  }
  if (upconv & 2) {
    if (int rc = up2c.run(bf.u1, bf.ups, bf.u2, V, 2 * H, 2 * W, 64, s)) return rc;
  } else {
    if (int rc = launch_resize_bilinear_ac(dtype, bf.u1, bf.ups, V, 2 * H, 2 * W, 256, 4 * H, 4 * W, 256, 0, s)) return rc;
    if (int rc = up2.run(bf.ups, bf.u2, V, 1, 4 * H, 4 * W, 64, nullptr, 0, nullptr, 0, s)) return rc;
  }
  fin.out_plain_f32 = feat_f32_only();
  if ((upconv & 4) && tail.ready())       // up_3 + final from the half-resolution tensor in one kernel: no up-sampled tensor, no z, no u3
    return tail.run(bf.u2, fin.out_plain_f32 ? (void*)bf.featf : bf.feat, fin.out_plain_f32 ? 1 : feat_f16() ? 2 : 0, V, 4 * H, 4 * W, s);
  if (int rc = launch_resize_bilinear_ac(dtype, bf.u2, bf.ups, V, 4 * H, 4 * W, 64, 8 * H, 8 * W, 64, 0, s)) return rc;
  // up_3 + final: one launch on the bf16 path (the 64-channel up_3 output then never reaches HBM: `u3` is not written)
  // bf16x3, default path: `final` writes the plain-fp32 feature map directly (no split-pair copy, no 6.6 GB conversion pass)
  fin.out_plain_f32 = feat_f32_only();
  if (int rc = up3.run_then_1x1(fin, bf.ups, bf.u3, 64, fin.out_plain_f32 ? (void*)bf.featf : bf.feat, 32, V, 1, 8 * H, 8 * W, fuse_final != 0, nullptr, s)) return rc;
  return 0;
}

int AdaPose::cost_volume(const Buffers& bf, int V, int B, const float* depths, hipStream_t s) const {
  const int S = img, D = n_depth, P = n_pts;
  const int Vh = view2_heads ? V : B;      // views whose probability volume is computed (partners are looked up among all V)
  const int Vc0 = chunk_views(V);
  const bool b16 = dtype_size(dtype) == 2;      // 16-bit storage: the depth-sweeping conv0, implicit-GEMM conv6 and the sparse tail exist for these
  // halo-tiled path (cost_impl >= 1): one launch per layer; conv0 optionally builds its input on the fly
  bool sweep_sparse = false;      // set per chunk below: the depth-sweeping conv0 walks the list of needed tiles
  auto tile = [&](int layer, const void* in, void* out, const void* res, int Vc, int Di, int Hi, int Wi, int Do, int Ho,
                  int Wo, bool transposed, int v0, const unsigned char* tmask = nullptr) -> int {
    const int li = layer == 10 ? 0 : layer;
    Conv3dTileDesc d;
    memset(&d, 0, sizeof(d));
    d.in = in; d.wgt = t3d[li].w; d.out = out; d.bias = t3d[li].bias; d.res = res;
    d.N = Vc; d.Di = Di; d.Hi = Hi; d.Wi = Wi; d.Do = Do; d.Ho = Ho; d.Wo = Wo;
    d.Dq = transposed ? Di : Do; d.Hq = transposed ? Hi : Ho; d.Wq = transposed ? Wi : Wo;
    d.Cout = t3d[li].Cout; d.relu = 1;
    d.feat = bf.feat; d.homog = bf.homog; d.depths = depths; d.v0 = v0; d.V = V; d.B = B;
    d.out_classmajor = layer == 9 ? 1 : 0;      // u11 is only gathered sparsely by the prob kernel
    d.tile_mask = tmask;
    d.tile_mask_stride = sparse_mask_bytes_per_view(S);
    if (layer == 10 && sweep_sparse) { d.tile_list = bf.sweep_list; d.tile_count = bf.sweep_count; }
    // profiler rows (prof.h): conv0 + fused warp on its own; bf16 layers one row each, f32 layers aggregated
    d.prof_variant = dtype == BF16X3 ? (layer == 10 ? 28 : 27) : layer == 10 ? 10 + (dtype != F32 ? 1 : 0) : (dtype != F32 ? 16 + layer : 8);
    d.algo_flops = 2.0 * Vc * (double)(transposed ? Di * Hi * Wi : Do * Ho * Wo) * t3d[li].Cout * 27.0 * t3d[li].Cin;
    // layer 10 (conv0 with the plane sweep fused in) reads the two feature maps of a pair, not the 32 x D x H x W volume it
    // never materialises: algorithmic bytes = features in + c0 out
    d.algo_bytes = ((layer == 10 ? (double)Vc * Hi * Wi * t3d[li].Cin : (double)Vc * Di * Hi * Wi * t3d[li].Cin) +
                    (double)Vc * Do * Ho * Wo * t3d[li].Cout * (res ? 2 : 1)) * (double)dtype_size(dtype);
    // debug flag 4096: the halo-tile conv0 instead of either depth-sweeping kernel, in every storage type (the runtime way out
    // should a compiler update bring the sweep kernels' hand-counted asm gathers out of step)
    if (layer == 10 && cost_impl == 3 && b16 && sweep_w && !(g_debug_flags & 4096)) {
      d.wgt = sweep_w;
      if (feat_f16()) { d.wgt = sweep_w_f16; d.feat_f16 = 1; }
      return launch_conv0_sweep(d, dtype, s);
    }
    if (layer == 10 && cost_impl == 3 && dtype == BF16X3 && sweep_w && !(g_debug_flags & 4096)) {
      d.wgt = sweep_w;
      d.feat = bf.featf;
      return launch_conv0_sweep_x3(d, s);
    }
    return launch_conv3d_tile(layer, dtype, d, s);
  };
  // Sparse cost regularisation (sparse_dec, with the sparse tail only): the tail reads the probability volume at the chosen pixels, so every
  // layer is needed only inside their dependency cones (prob_sparse.hip: sparse_mask_kernel) — conv1 .. conv5, conv7 / conv9 skip the
  // other tiles, the depth-sweeping conv0 walks the list of needed ones.  conv6 stays dense (a 0.3 ms GEMM): what it computes from
  // unwritten input tiles is never read by anything that is read.
  const bool sparse_ok = sparse_active();
  for (int v0 = 0; norm_mode == 0 && cost_impl >= 1 && v0 < Vh; v0 += Vc0) {
    const int Vc = Vh - v0 < Vc0 ? Vh - v0 : Vc0;
    const unsigned char* mk[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    sweep_sparse = false;
    if (sparse_ok) {
      int td, th, tw;
      for (int L : {1, 2, 3, 4, 5, 7, 8}) RGBM_REQUIRE(!conv3d_tile_dims(L, dtype, &td, &th, &tw) && th == 8 && tw == 8, "sparse cost regularisation: 8 x 8 tiles expected");
      if (int rc = launch_sparse_masks(bf.choose, v0, Vc, P, S, bf.masks, bf.sweep_list, bf.sweep_count, s)) return rc;
      for (int L : {7, 8}) mk[L] = bf.masks + sparse_mask_offset(S, L);
      if (sparse_dec >= 2) {
        for (int L : {1, 2, 3, 4, 5}) mk[L] = bf.masks + sparse_mask_offset(S, L);
        sweep_sparse = true;
      }
    }
    if (cost_impl == 1) {
      if (int rc = launch_build_volume(dtype, bf.feat, bf.homog, depths, bf.vol, v0, Vc, V, B, D, S, S, s)) return rc;
      if (int rc = tile(0, bf.vol, bf.c[0], nullptr, Vc, D, S, S, D, S, S, false, v0)) return rc;
    } else {
      if (int rc = tile(10, nullptr, bf.c[0], nullptr, Vc, D, S, S, D, S, S, false, v0)) return rc;
    }
    if (int rc = tile(1, bf.c[0], bf.c[1], nullptr, Vc, D, S, S, D / 2, S / 2, S / 2, false, v0, mk[1])) return rc;
    if (int rc = tile(2, bf.c[1], bf.c[2], nullptr, Vc, D / 2, S / 2, S / 2, D / 2, S / 2, S / 2, false, v0, mk[2])) return rc;
    if (int rc = tile(3, bf.c[2], bf.c[3], nullptr, Vc, D / 2, S / 2, S / 2, D / 4, S / 4, S / 4, false, v0, mk[3])) return rc;
    if (int rc = tile(4, bf.c[3], bf.c[4], nullptr, Vc, D / 4, S / 4, S / 4, D / 4, S / 4, S / 4, false, v0, mk[4])) return rc;
    if (int rc = tile(5, bf.c[4], bf.c[5], nullptr, Vc, D / 4, S / 4, S / 4, D / 8, S / 8, S / 8, false, v0, mk[5])) return rc;
    if (cost_impl == 3 && (b16 || dtype == BF16X3) && igemm_conv6) {
      // conv6 (64 -> 64, K = 27 x 64): a plain GEMM shape, 2.7x faster on the role-specialised implicit-GEMM kernel
      if (int rc = c3d[6].run(bf.c[5], bf.c[6], Vc, D / 8, S / 8, S / 8, 64, nullptr, RES_NONE, nullptr, 0, s)) return rc;
    } else {
      if (int rc = tile(6, bf.c[5], bf.c[6], nullptr, Vc, D / 8, S / 8, S / 8, D / 8, S / 8, S / 8, false, v0)) return rc;
    }
    if (int rc = tile(7, bf.c[6], bf.u7, bf.c[4], Vc, D / 8, S / 8, S / 8, D / 4, S / 4, S / 4, true, v0, mk[7])) return rc;
    if (int rc = tile(8, bf.u7, bf.u9, bf.c[2], Vc, D / 4, S / 4, S / 4, D / 2, S / 2, S / 2, true, v0, mk[8])) return rc;
    if (cost_impl == 3 && (b16 || (dtype == BF16X3 && w11_x3)) && sparse_tail) {
      // conv11 + skip + prob conv + softmax + depth only on the 3x3 neighbourhoods of the chosen pixels (prob_sparse.hip)
      if (int rc = launch_prob_sparse(bf.u9, bf.c[0], dtype == BF16X3 ? w11_x3 : t3d[9].w, t3d[9].bias, wprob, bf.choose, depths, bf.prob,
                                      bf.depth, v0, Vc, B, P, D, S, S, dtype, s, w11_taps)) return rc;
      continue;
    }
    if (int rc = tile(9, bf.u9, bf.u11, bf.c[0], Vc, D / 2, S / 2, S / 2, D, S, S, true, v0)) return rc;
    if (int rc = launch_prob_softmax_depth(dtype, bf.u11, wprob, bf.choose, depths, bf.prob, bf.depth, v0, Vc, B, P, D, S, S, 1, s)) return rc;
  }
  // norm_mode 1: every layer = un-normalised conv (generic implicit GEMM) -> per-view batch statistics -> normalise + ReLU (+ skip)
  for (int v0 = 0; norm_mode == 1 && v0 < Vh; v0 += Vc0) {
    const int Vc = Vh - v0 < Vc0 ? Vh - v0 : Vc0;
    if (int rc = launch_build_volume(dtype, bf.feat, bf.homog, depths, bf.vol, v0, Vc, V, B, D, S, S, s)) return rc;
    auto layer = [&](int i, const void* in, void* out, const void* res, int Di, int Hi, int Wi, int C) -> int {
      const ConvLayer& L = i >= 7 ? dc_raw[i - 7] : c3d_raw[i];
      int Do, Ho, Wo;
      L.out_dims(Di, Hi, Wi, Do, Ho, Wo);
      if (int rc = L.run(in, out, Vc, Di, Hi, Wi, C, nullptr, RES_NONE, nullptr, 0, s)) return rc;
      return launch_bn_per_sample(dtype, out, res, bn_gamma[i], bn_beta[i], bf.bn_scratch, Vc, (long long)Do * Ho * Wo, C, 1, s);
    };
    if (int rc = layer(0, bf.vol, bf.c[0], nullptr, D, S, S, 8)) return rc;
    if (int rc = layer(1, bf.c[0], bf.c[1], nullptr, D, S, S, 16)) return rc;
    if (int rc = layer(2, bf.c[1], bf.c[2], nullptr, D / 2, S / 2, S / 2, 16)) return rc;
    if (int rc = layer(3, bf.c[2], bf.c[3], nullptr, D / 2, S / 2, S / 2, 32)) return rc;
    if (int rc = layer(4, bf.c[3], bf.c[4], nullptr, D / 4, S / 4, S / 4, 32)) return rc;
    if (int rc = layer(5, bf.c[4], bf.c[5], nullptr, D / 4, S / 4, S / 4, 64)) return rc;
    if (int rc = layer(6, bf.c[5], bf.c[6], nullptr, D / 8, S / 8, S / 8, 64)) return rc;
    if (int rc = layer(7, bf.c[6], bf.u7, bf.c[4], D / 8, S / 8, S / 8, 32)) return rc;
    if (int rc = layer(8, bf.u7, bf.u9, bf.c[2], D / 4, S / 4, S / 4, 16)) return rc;
    if (int rc = layer(9, bf.u9, bf.u11, bf.c[0], D / 2, S / 2, S / 2, 8)) return rc;
    if (int rc = launch_prob_softmax_depth(dtype, bf.u11, wprob, bf.choose, depths, bf.prob, bf.depth, v0, Vc, B, P, D, S, S, 0, s)) return rc;
  }
  if (norm_mode == 1) return 0;
  for (int v0 = 0; cost_impl == 0 && v0 < Vh; v0 += Vc0) {
    const int Vc = Vh - v0 < Vc0 ? Vh - v0 : Vc0;
    if (int rc = launch_build_volume(dtype, bf.feat, bf.homog, depths, bf.vol, v0, Vc, V, B, D, S, S, s)) return rc;
    if (int rc = c3d[0].run(bf.vol, bf.c[0], Vc, D, S, S, 8, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = c3d[1].run(bf.c[0], bf.c[1], Vc, D, S, S, 16, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = c3d[2].run(bf.c[1], bf.c[2], Vc, D / 2, S / 2, S / 2, 16, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = c3d[3].run(bf.c[2], bf.c[3], Vc, D / 2, S / 2, S / 2, 32, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = c3d[4].run(bf.c[3], bf.c[4], Vc, D / 4, S / 4, S / 4, 32, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = c3d[5].run(bf.c[4], bf.c[5], Vc, D / 4, S / 4, S / 4, 64, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = c3d[6].run(bf.c[5], bf.c[6], Vc, D / 8, S / 8, S / 8, 64, nullptr, 0, nullptr, 0, s)) return rc;
    // skip adds are post-ReLU (network_v5.py:287-289)
    if (int rc = dc[0].run(bf.c[6], bf.u7, Vc, D / 8, S / 8, S / 8, 32, bf.c[4], RES_POST_ACT, nullptr, 0, s)) return rc;
    if (int rc = dc[1].run(bf.u7, bf.u9, Vc, D / 4, S / 4, S / 4, 16, bf.c[2], RES_POST_ACT, nullptr, 0, s)) return rc;
    if (int rc = dc[2].run(bf.u9, bf.u11, Vc, D / 2, S / 2, S / 2, 8, bf.c[0], RES_POST_ACT, nullptr, 0, s)) return rc;
    if (int rc = launch_prob_softmax_depth(dtype, bf.u11, wprob, bf.choose, depths, bf.prob, bf.depth, v0, Vc, B, P, D, S, S, 0, s)) return rc;
  }
  return 0;
}

int AdaPose::forward(int B, const float* img1, const float* img2, const int* choose1, const int* choose2, const float* P1,
                     const float* P2, const float* depths, void* workspace, size_t workspace_size, const Outputs& out,
                     hipStream_t s, int stop_after) const {
  RGBM_REQUIRE(B > 0, "batch");
  RGBM_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  RGBM_REQUIRE(n_depth % 8 == 0 && img % 8 == 0, "depth/img must be multiples of 8");
  const size_t need = workspace_bytes(B);
  RGBM_REQUIRE(workspace_size >= need, "workspace too small: need " + std::to_string(need));
  Arena A(workspace, workspace_size);
  Buffers bf;
  plan(B, A, bf);
  const int V = 2 * B, P = n_pts, S = img, D = n_depth;
  const size_t VP = (size_t)V * P;

  // ---- stage inputs: views = [view1 batch ; view2 batch] ----
  if (int rc = launch_stage_in(P1, P2, choose1, choose2, bf.Pviews, bf.choose, B, P, s)) return rc;
  if (int rc = pspnet(bf, V, img1, img2, s)) return rc;
  if (int rc = launch_homography(bf.Pviews, bf.homog, V, B, s)) return rc;
  // bf16x3 nets: everything that GATHERS from the feature map (plane sweep, point heads) reads a plain fp32 copy of it
  const void* featg = bf.feat;
  int fdt = dtype;
  if (dtype == BF16X3) {
    if (!feat_f32_only())
      if (int rc = launch_bx3_to_f32(bf.feat, bf.featf, (long long)V * S * S * 32, s)) return rc;
    featg = bf.featf; fdt = F32;
  }
  if (feat_f16()) fdt = F16;
  if (stop_after == 1) return 0;

  const int Vh = view2_heads ? V : B;      // views that get heads: both crops of every pose, or the view-1 crops only (option view2_heads)
  // ---- per-point NOCS branch (network_v5.py:432-444) ----
  if (pmlp_table != nullptr && !(g_debug_flags & 2048) && ((long long)Vh * P) % 64 == 0) {
    // gather + the six layers in one launch (head_kernels.hip): a wave carries 16 points through the whole branch, weights and activations in LDS
    PointMlpDesc pd{};
    pd.table = pmlp_table; pd.feat = featg; pd.choose = bf.choose; pd.nocs4 = bf.nocs4; pd.pf = bf.PF96 + 32; pd.ldpf = 96; pd.P = P; pd.HW = S * S;
    pd.N = (long long)Vh * P;
    if (int rc = launch_point_mlp(fdt, pd, s)) return rc;
  } else {
    if (int rc = launch_gather_points(fdt, featg, bf.choose, bf.X0, Vh, P, S * S, 32, s)) return rc;
    if (int rc = inst.run(bf.X0, bf.X1, Vh, 1, 1, P, 64, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = nh[0].run(bf.X1, bf.H128, Vh, 1, 1, P, 128, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = nh[1].run(bf.H128, bf.H64, Vh, 1, 1, P, 64, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = nh[2].run(bf.H64, bf.nocs4, Vh, 1, 1, P, 4, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = npm[0].run(bf.nocs4, bf.N32, Vh, 1, 1, P, 32, nullptr, 0, nullptr, 0, s)) return rc;
    if (int rc = npm[1].run(bf.N32, bf.PF96 + 32, Vh, 1, 1, P, 96, nullptr, 0, nullptr, 0, s)) return rc;
  }

  // ---- plane-sweep cost volume -> probability at the sampled pixels -> depth ----
  if (int rc = cost_volume(bf, V, B, depths, s)) return rc;
  if (stop_after == 2) return 0;

  // ---- depth-guided fusion + pose regression (network_v5.py:457-508) ----
  if (int rc = launch_fuse_points(fdt, featg, bf.homog, depths, bf.choose, bf.prob, bf.PF96, V, B, P, D, S, S, 96, 0, s, Vh)) return rc;
  // bf16 nets: the four big per-point layers of the pose MLP run in fp16 storage (fp32 they took 1.8 ms per 512 views at
  // 78 TFLOP/s on the fp32 matrix path); PF96 is produced in fp32 by its two writers and converted once
  const int pdt = pose_dtype();
  const void* pf_in = bf.PF96;
  if (pdt == F16) {
    if (int rc = launch_f32_to_f16(bf.PF96, bf.PF96h, (long long)Vh * P * 96, s)) return rc;
    pf_in = bf.PF96h;
  } else if (pdt == BF16X3) {
    if (int rc = launch_f32_to_bx3(bf.PF96, bf.PF96, (long long)Vh * P * 96, s)) return rc;      // in place: same 4-byte slots
  }
  if (int rc = pm1[0].run(pf_in, bf.Q128a, Vh, 1, 1, P, 128, nullptr, 0, nullptr, 0, s)) return rc;
  if (int rc = pm1[1].run(bf.Q128a, bf.Q128b, Vh, 1, 1, P, 128, nullptr, 0, nullptr, 0, s)) return rc;
  // the two means over a view's points are finished by their consumers (the slices' partial sums in Q128a / G256a: free at that point)
  if (int rc = launch_mean_points_partial(pdt, bf.Q128b, (float*)bf.Q128a, Vh, P, 128, s)) return rc;
  if (int rc = launch_view_linear_mean((const float*)bf.Q128a, P, bf.glob, pm2_0_wfull, pm2_0_bias, bf.vbias, Vh, 128, 256, 256, 128, 0, s)) return rc;
  if (int rc = pm2[0].run(bf.Q128b, bf.G256a, Vh, 1, 1, P, 256, nullptr, 0, bf.vbias, 256, s)) return rc;
  if (int rc = pm2[1].run(bf.G256a, bf.G256b, Vh, 1, 1, P, 256, nullptr, 0, nullptr, 0, s)) return rc;
  if (int rc = launch_mean_points_partial(pdt, bf.G256b, (float*)bf.G256a, Vh, P, 256, s)) return rc;
  float* hout[3] = {bf.r6, bf.tv, bf.sv};
  const int hdim[3] = {6, 3, 3};
  if (int rc = launch_pose_heads_mean((const float*)bf.G256a, P, bf.pf2, bf.R, head_w, head_b, hout, hdim, Vh, s)) return rc;      // + Ortho6d -> R

  // ---- outputs (fp32, reference shapes); view2_heads = 0: the view-2 outputs are filled with NaN, so that a consumer of one fails
  // loudly instead of reading stale numbers ----
  {
    float* const on[2] = {out.nocs1, out.nocs2};
    float* const od[2] = {out.depth1, out.depth2};
    float* const orr[2] = {out.r1, out.r2};
    float* const ot[2] = {out.t1, out.t2};
    float* const os[2] = {out.s1, out.s2};
    if (int rc = launch_stage_out(bf.nocs4, bf.depth, bf.R, bf.tv, bf.sv, on, od, orr, ot, os, B, P, view2_heads, s)) return rc;
  }
  (void)VP;
  return 0;
}

}  // namespace rgbm
