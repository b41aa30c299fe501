// extern "C" boundary of librgbm_hip.so — see include/rgbm.h for the contract and the reference
// interfaces each entry point replaces.
#include "../../include/rgbm.h"

#include <string.h>

#include <string>
#include <vector>

#include "adapose.h"
#include "control.h"
#include "prof.h"

using namespace rgbm;

// One captured forward: the launch sequence of AdaPose::forward for a fixed batch size and fixed device pointers, replayed with a
// single hipGraphLaunch (small batches are launch-bound: ~150 launches of a few microseconds each per forward).
struct ForwardGraph {
  int B = 0;
  const void* in[7] = {nullptr};
  void* ws = nullptr;
  size_t ws_bytes = 0;
  rgbm_adapose_out out;
  int opt_version = 0;
  int tuning_version = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  size_t nodes = 0;
  unsigned long long last_use = 0;
};

struct rgbm_adapose {
  AdaPose net;
  int device;
  int opt_version = 0;                 // bumped by set_option / set_chunk: captured graphs of older settings are dropped
  unsigned long long tick = 0;
  std::vector<ForwardGraph> graphs;    // at most kMaxGraphs, least recently used evicted
};
static const size_t kMaxGraphs = 8;

static void drop_graph(ForwardGraph& g) {
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  g.exec = nullptr; g.graph = nullptr;
}


extern "C" {

int rgbm_version(void) { return RGBM_VERSION; }
const char* rgbm_last_error(void) { return last_error_cstr(); }

int rgbm_adapose_create(rgbm_adapose_t** h, int device, const rgbm_weight_desc* w, int n_w, int dtype, int norm_mode) {
  RGBM_REQUIRE(h != nullptr && w != nullptr && n_w > 0, "create arguments");
  RGBM_REQUIRE(dtype == RGBM_F32 || dtype == RGBM_BF16 || dtype == RGBM_F16 || dtype == RGBM_BF16X3, "dtype must be 0 (fp32), 1 (bf16), 2 (fp16) or 3 (bf16x3)");
  RGBM_REQUIRE(norm_mode == 0 || norm_mode == 1, "norm_mode: 0 = eval-mode (folded) BatchNorm3d, 1 = per-sample statistics");
  RGBM_CHECK_HIP(hipSetDevice(device));
  StateDict sd;
  for (int i = 0; i < n_w; ++i) {
    std::string name = w[i].name;
    if (name.rfind("module.", 0) == 0) name = name.substr(7);
    HostTensor t;
    t.data = w[i].data;
    for (int d = 0; d < w[i].ndim; ++d) t.shape.push_back(w[i].shape[d]);
    sd[name] = t;
  }
  rgbm_adapose* obj = new rgbm_adapose();
  obj->device = device;
  int rc = obj->net.create(sd, dtype, norm_mode);
  if (rc) { obj->net.destroy(); delete obj; return rc; }
  *h = obj;
  return 0;
}

int rgbm_adapose_destroy(rgbm_adapose_t* h) {
  if (!h) return 0;
  for (auto& g : h->graphs) drop_graph(g);
  h->net.destroy();
  delete h;
  return 0;
}

int rgbm_adapose_set_chunk(rgbm_adapose_t* h, int max_chunk_views) {
  RGBM_REQUIRE(h && max_chunk_views > 0, "set_chunk arguments");
  h->net.max_chunk = max_chunk_views;
  ++h->opt_version;
  return 0;
}

int rgbm_adapose_set_option(rgbm_adapose_t* h, const char* key, int value) {
  RGBM_REQUIRE(h && key, "set_option arguments");
  const std::string k = key;
  if (k == "max_chunk") { RGBM_REQUIRE(value > 0, "max_chunk"); h->net.max_chunk = value; }
  else if (k == "igemm_conv6") { RGBM_REQUIRE(value == 0 || value == 1, "igemm_conv6"); h->net.igemm_conv6 = value; }
  else if (k == "fuse_final") { RGBM_REQUIRE(value == 0 || value == 1, "fuse_final"); h->net.fuse_final = value; }
  else if (k == "sparse_tail") { RGBM_REQUIRE(value == 0 || value == 1, "sparse_tail"); h->net.sparse_tail = value; }
  else if (k == "upconv") { RGBM_REQUIRE(value >= 0 && value <= 7, "upconv"); h->net.upconv = value; }
  else if (k == "sparse_dec") { RGBM_REQUIRE(value >= 0 && value <= 2, "sparse_dec"); h->net.sparse_dec = value; }
  else if (k == "stem") { RGBM_REQUIRE(value == 0 || value == 1, "stem"); h->net.stem = value; }
  else if (k == "view2_heads") { RGBM_REQUIRE(value == 0 || value == 1, "view2_heads"); h->net.view2_heads = value; }
  else if (k == "sweep_f16") { RGBM_REQUIRE(value == 0 || value == 1, "sweep_f16"); h->net.sweep_f16 = value; }
  else if (k == "cost_impl") { RGBM_REQUIRE(value >= 0 && value <= 3, "cost_impl"); h->net.cost_impl = value; }
  else { set_error("unknown option " + k); return -1; }
  ++h->opt_version;
  return 0;
}

int rgbm_adapose_workspace_bytes(rgbm_adapose_t* h, int B, size_t* bytes) {
  RGBM_REQUIRE(h && bytes && B > 0, "workspace_bytes arguments");
  *bytes = h->net.workspace_bytes(B);
  return 0;
}

static AdaPose::Outputs to_out(const rgbm_adapose_out* o) {
  AdaPose::Outputs r;
  r.nocs1 = o->view1_nocs; r.nocs2 = o->view2_nocs; r.depth1 = o->view1_depth; r.depth2 = o->view2_depth;
  r.r1 = o->view1_r; r.r2 = o->view2_r; r.t1 = o->view1_t; r.t2 = o->view2_t; r.s1 = o->view1_s; r.s2 = o->view2_s;
  return r;
}

int rgbm_adapose_forward_ex(rgbm_adapose_t* h, int B, const float* img1, const float* img2, const int32_t* choose1,
                            const int32_t* choose2, const float* P1, const float* P2, const float* depths, void* workspace,
                            size_t workspace_bytes, const rgbm_adapose_out* out, int stop_after, void* stream) {
  RGBM_REQUIRE(h && img1 && img2 && choose1 && choose2 && P1 && P2 && depths && workspace && out, "forward arguments");
  return h->net.forward(B, img1, img2, choose1, choose2, P1, P2, depths, workspace, workspace_bytes, to_out(out),
                        (hipStream_t)stream, stop_after);
}

int rgbm_adapose_forward(rgbm_adapose_t* h, int B, const float* img1, const float* img2, const int32_t* choose1,
                         const int32_t* choose2, const float* P1, const float* P2, const float* depths, void* workspace,
                         size_t workspace_bytes, const rgbm_adapose_out* out, void* stream) {
  return rgbm_adapose_forward_ex(h, B, img1, img2, choose1, choose2, P1, P2, depths, workspace, workspace_bytes, out, 0, stream);
}

int rgbm_adapose_graph_clear(rgbm_adapose_t* h) {
  RGBM_REQUIRE(h, "graph_clear arguments");
  for (auto& g : h->graphs) drop_graph(g);
  h->graphs.clear();
  return 0;
}

int rgbm_adapose_forward_graph(rgbm_adapose_t* h, int B, const float* img1, const float* img2, const int32_t* choose1,
                               const int32_t* choose2, const float* P1, const float* P2, const float* depths, void* workspace,
                               size_t workspace_bytes, const rgbm_adapose_out* out, void* stream, int32_t* n_nodes, int32_t* captured) {
  RGBM_REQUIRE(h && img1 && img2 && choose1 && choose2 && P1 && P2 && depths && workspace && out, "forward_graph arguments");
  hipStream_t s = (hipStream_t)stream;
  if (n_nodes) *n_nodes = 0;
  if (captured) *captured = 0;
  // the in-library profiler records events around launches: not capturable, and a timing run wants the launches themselves
  if (prof_enabled()) {
    if (captured) *captured = -1;
    return h->net.forward(B, img1, img2, choose1, choose2, P1, P2, depths, workspace, workspace_bytes, to_out(out), s, 0);
  }
  RGBM_REQUIRE(s != nullptr, "forward_graph: stream capture is not allowed on the default (null) stream; pass a created stream");
  const void* in[7] = {img1, img2, choose1, choose2, P1, P2, depths};
  ForwardGraph* hit = nullptr;
  for (size_t i = 0; i < h->graphs.size();) {
    ForwardGraph& g = h->graphs[i];
    if (g.opt_version != h->opt_version || g.tuning_version != g_tuning_version) {      // settings changed since the capture
      drop_graph(g);
      h->graphs.erase(h->graphs.begin() + i);
      continue;
    }
    if (g.B == B && !memcmp(g.in, in, sizeof(in)) && g.ws == workspace && g.ws_bytes == workspace_bytes &&
        !memcmp(&g.out, out, sizeof(*out))) hit = &g;
    ++i;
  }
  if (!hit) {
    // an eager forward first: it sets the dynamic-LDS attributes of the kernels this shape uses and loads their code objects (neither
    // may happen inside a capture), and reports argument errors with their own messages
    if (int rc = h->net.forward(B, img1, img2, choose1, choose2, P1, P2, depths, workspace, workspace_bytes, to_out(out), s, 0)) return rc;
    ForwardGraph g;
    g.B = B; memcpy(g.in, in, sizeof(in)); g.ws = workspace; g.ws_bytes = workspace_bytes; g.out = *out;
    g.opt_version = h->opt_version; g.tuning_version = g_tuning_version;
    RGBM_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = h->net.forward(B, img1, img2, choose1, choose2, P1, P2, depths, workspace, workspace_bytes, to_out(out), s, 0);
    const hipError_t ee = hipStreamEndCapture(s, &g.graph);
    if (rc) { if (g.graph) (void)hipGraphDestroy(g.graph); return rc; }
    if (ee != hipSuccess || !g.graph) { set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(ee)); return -2; }
    if (hipGraphGetNodes(g.graph, nullptr, &g.nodes) != hipSuccess) g.nodes = 0;
    const hipError_t ei = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) { (void)hipGraphDestroy(g.graph); set_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); return -2; }
    if (h->graphs.size() >= kMaxGraphs) {
      size_t lru = 0;
      for (size_t i = 1; i < h->graphs.size(); ++i) if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
      drop_graph(h->graphs[lru]);
      h->graphs.erase(h->graphs.begin() + lru);
    }
    h->graphs.push_back(g);
    hit = &h->graphs.back();
    if (captured) *captured = 1;
  }
  hit->last_use = ++h->tick;
  if (n_nodes) *n_nodes = (int32_t)hit->nodes;
  RGBM_CHECK_HIP(hipGraphLaunch(hit->exec, s));
  return 0;
}

int rgbm_adapose_fetch(rgbm_adapose_t* h, int B, void* workspace, const char* name, float* out_dev, size_t capacity,
                       size_t* n_elems, void* stream) {
  RGBM_REQUIRE(h && workspace && name && out_dev && n_elems, "fetch arguments");
  const AdaPose& n = h->net;
  Arena A(workspace, 0);
  AdaPose::Buffers bf;
  n.plan(B, A, bf);
  const size_t V = 2 * (size_t)B, S = n.img, P = n.n_pts, D = n.n_depth;
  const size_t Vc = n.chunk_views((int)V);
  const std::string nm = name;
  const void* src = nullptr; size_t cnt = 0; int dt = n.dtype;
  if (nm == "imgpad") { src = bf.imgpad; cnt = V * S * S * n.img_cpad; }
  else if (nm == "conv1") { src = bf.c1; cnt = V * (S / 2) * (S / 2) * 64; }
  else if (nm == "layer4") { src = bf.lb[n.last_f_index]; cnt = V * (S / 8) * (S / 8) * 512; }
  else if (nm == "cat") { src = bf.cat; cnt = V * (S / 8) * (S / 8) * 1024; }
  else if (nm == "u1") { src = bf.u1; cnt = V * (S / 4) * (S / 4) * 256; }
  else if (nm == "u2") { src = bf.u2; cnt = V * (S / 2) * (S / 2) * 64; }
  else if (nm == "u3") { src = bf.u3; cnt = V * S * S * 64; }
  else if (nm == "feat") { src = bf.feat; cnt = V * S * S * 32; if (n.feat_f32_only()) { src = bf.featf; dt = F32; } else if (n.feat_f16()) dt = F16; }
  else if (nm == "vol") { src = bf.vol; cnt = Vc * D * S * S * 32; }
  else if (nm == "c0") { src = bf.c[0]; cnt = Vc * D * S * S * 8; }
  else if (nm == "c2") { src = bf.c[2]; cnt = Vc * (D / 2) * (S / 2) * (S / 2) * 16; }
  else if (nm == "c4") { src = bf.c[4]; cnt = Vc * (D / 4) * (S / 4) * (S / 4) * 32; }
  else if (nm == "c6") { src = bf.c[6]; cnt = Vc * (D / 8) * (S / 8) * (S / 8) * 64; }
  else if (nm == "u7") { src = bf.u7; cnt = Vc * (D / 4) * (S / 4) * (S / 4) * 32; }
  else if (nm == "u9") { src = bf.u9; cnt = Vc * (D / 2) * (S / 2) * (S / 2) * 16; }
  else if (nm == "u11") { src = bf.u11; cnt = Vc * D * S * S * 8; }
  else if (nm == "homog") { src = bf.homog; cnt = V * 12; dt = F32; }
  else if (nm == "prob") { src = bf.prob; cnt = V * P * D; dt = F32; }
  // pf96: after a forward of a split-pair net the buffer holds hi / lo pairs (adapose.cpp converts it in place for the pose MLP)
  else if (nm == "pf96") { src = bf.PF96; cnt = V * P * 96; dt = n.pose_dtype() == BF16X3 ? BF16X3 : F32; }
  else if (nm == "pf2") { src = bf.pf2; cnt = V * 256; dt = F32; }
  else if (nm == "r6") { src = bf.r6; cnt = V * 6; dt = F32; }
  RGBM_REQUIRE(src != nullptr, "unknown intermediate name: " + nm);
  // with sparse cost regularisation c0..c5 / u7 / u9 are written only inside the chosen pixels' dependency cones (and c6 is computed
  // from them): the rest of these tensors is whatever the workspace held, so a tap of them is refused instead of returned partly stale
  const bool tap3d = nm == "c0" || nm == "c2" || nm == "c4" || nm == "c6" || nm == "u7" || nm == "u9";
  RGBM_REQUIRE(!(tap3d && n.sparse_active()), "tap '" + nm + "' needs a dense cost regularisation: set option sparse_dec = 0 before the forward "
               "(the default computes it only inside the chosen pixels' dependency cones)");
  *n_elems = cnt;
  RGBM_REQUIRE(cnt <= capacity, "fetch capacity too small");
  return launch_to_f32(dt, src, out_dev, (long long)cnt, (hipStream_t)stream);
}

int rgbm_adapose_postprocess(int B, int P, int img_size, const float* nocs1, const float* depth1, const float* r1,
                             const int32_t* choose1, const double* Kcrop, const double* E1, double* bbox_world, double* ts,
                             int32_t* valid, void* stream) {
  RGBM_REQUIRE(B > 0 && nocs1 && depth1 && r1 && choose1 && Kcrop && E1 && bbox_world && ts && valid, "postprocess arguments");
  return launch_postprocess(nocs1, depth1, r1, choose1, Kcrop, E1, bbox_world, ts, valid, B, P, img_size, (hipStream_t)stream);
}

int rgbm_adapose_postprocess_scratch_bytes(int B, size_t* bytes) {
  RGBM_REQUIRE(B > 0 && bytes, "postprocess_scratch_bytes arguments");
  *bytes = postprocess_slices(B) > 1 ? postprocess_scratch_bytes(B) : 0;
  return 0;
}

int rgbm_adapose_postprocess_ws(int B, int P, int img_size, const float* nocs1, const float* depth1, const float* r1,
                                const int32_t* choose1, const double* Kcrop, const double* E1, double* bbox_world, double* ts,
                                int32_t* valid, void* scratch, size_t scratch_bytes, void* stream) {
  RGBM_REQUIRE(B > 0 && nocs1 && depth1 && r1 && choose1 && Kcrop && E1 && bbox_world && ts && valid, "postprocess arguments");
  return launch_postprocess(nocs1, depth1, r1, choose1, Kcrop, E1, bbox_world, ts, valid, B, P, img_size, (hipStream_t)stream,
                            scratch, scratch_bytes);
}

int rgbm_adapose_postprocess_ransac(int B, int P, int img_size, uint32_t seed, const float* nocs1, const float* depth1,
                                    const int32_t* choose1, const double* Kcrop, const double* E1, double* bbox_world, double* srt,
                                    int32_t* valid, void* stream) {
  return launch_umeyama_ransac(nocs1, depth1, choose1, Kcrop, E1, bbox_world, srt, valid, B, P, img_size, seed, (hipStream_t)stream);
}

int rgbm_gae(int T, int N, const float* rewards, const uint8_t* dones, const float* values, const float* last_values,
             float gamma, float lam, float* returns, float* adv, double* sums, void* stream) {
  RGBM_REQUIRE(rewards && dones && values && last_values && returns && adv && sums, "gae arguments");
  return launch_gae(T, N, rewards, dones, values, last_values, gamma, lam, returns, adv, sums, (hipStream_t)stream);
}

int rgbm_adv_normalise(int64_t n_local, float* adv, const double* sums, double count_total, void* stream) {
  RGBM_REQUIRE(adv && sums && n_local > 0 && count_total > 1.0, "adv_normalise arguments");
  return launch_adv_normalise(n_local, adv, sums, count_total, (hipStream_t)stream);
}

int rgbm_conv_nd(int dtype, const void* in_dev, int N, int D, int H, int W, int Cin, int Cin_pad, const float* w_host,
                 int Cout, int Cout_pad, int KD, int KH, int KW, int stride_d, int stride_hw, int pad_d, int pad_hw,
                 int dil_hw, int transposed, const float* bias_host, const float* bn_scale_host, const float* bn_shift_host,
                 const void* res_dev, int res_mode, int act, float slope, void* out_dev, void* stream) {
  RGBM_REQUIRE(in_dev && w_host && out_dev, "conv arguments");
  ConvGeom g;
  g.Cin = Cin; g.Cout = Cout; g.KD = KD; g.KH = KH; g.KW = KW;
  g.sd = stride_d; g.sh = g.sw = stride_hw; g.pd = pad_d; g.ph = g.pw = pad_hw;
  g.dild = 1; g.dilh = g.dilw = dil_hw; g.transposed = transposed != 0; g.act = act; g.slope = slope;
  ConvLayer L;
  int rc = L.init(dtype, g, w_host, bias_host, bn_scale_host, bn_shift_host, Cin_pad, Cout_pad);
  if (!rc) rc = L.run(in_dev, out_dev, N, D, H, W, Cout_pad, res_dev, res_mode, nullptr, 0, (hipStream_t)stream);
  if (!rc) { hipError_t e = hipStreamSynchronize((hipStream_t)stream); if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = -2; } }
  L.destroy();
  return rc;
}

int rgbm_upsample_conv3x3(int dtype, const void* in_dev, int V, int h, int w, int Cin, const float* w_host, int Cout,
                          const float* bias_host, int act, float slope, void* z_scratch_dev, void* out_dev, void* stream) {
  RGBM_REQUIRE(in_dev && w_host && z_scratch_dev && out_dev, "upsample_conv3x3 arguments");
  UpConvLayer L;
  int rc = L.init(dtype, Cin, Cout, w_host, bias_host, act, slope);
  if (!rc) rc = L.run(in_dev, z_scratch_dev, out_dev, V, h, w, Cout, (hipStream_t)stream);
  if (!rc) { hipError_t e = hipStreamSynchronize((hipStream_t)stream); if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = -2; } }
  L.destroy();
  return rc;
}

int rgbm_upsample_conv3x3_final(int dtype, const void* in_dev, int V, int h, int w, const float* w3_host, const float* b3_host,
                                float slope, const float* wf_host, const float* bf_host, void* out_dev, int out_f32, void* stream) {
  RGBM_REQUIRE(in_dev && w3_host && b3_host && wf_host && bf_host && out_dev, "upsample_conv3x3_final arguments");
  UpConvFinal L;
  int rc = L.init(dtype, w3_host, b3_host, slope, wf_host, bf_host);
  if (!rc) rc = L.run(in_dev, out_dev, out_f32, V, h, w, (hipStream_t)stream);
  if (!rc) { hipError_t e = hipStreamSynchronize((hipStream_t)stream); if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = -2; } }
  L.destroy();
  return rc;
}

int rgbm_stem(int dtype, const float* img1_dev, const float* img2_dev, const float* w_host, void* out_dev, int B, int S, void* stream) {
  RGBM_REQUIRE(img1_dev && img2_dev && w_host && out_dev && B > 0, "stem arguments");
  std::vector<float> pk;
  stem_pack(w_host, pk);
  void* wd = nullptr;
  int rc = upload_packed(pk, dtype, &wd);
  if (!rc) rc = launch_stem(dtype, img1_dev, img2_dev, wd, out_dev, B, 2 * B, S, (hipStream_t)stream);
  if (!rc) { hipError_t e = hipStreamSynchronize((hipStream_t)stream); if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = -2; } }
  if (wd) (void)hipFree(wd);
  return rc;
}

int rgbm_maxpool3x3s2(int dtype, const void* in_dev, void* out_dev, int V, int H, int W, int C, void* stream) {
  return launch_maxpool3x3s2(dtype, in_dev, out_dev, V, H, W, C, (hipStream_t)stream);
}
int rgbm_resize_bilinear_ac(int dtype, const void* in_dev, void* out_dev, int V, int Hs, int Ws, int C, int Ho, int Wo,
                            void* stream) {
  return launch_resize_bilinear_ac(dtype, in_dev, out_dev, V, Hs, Ws, C, Ho, Wo, C, 0, (hipStream_t)stream);
}
int rgbm_adaptive_avgpool(int dtype, const void* in_dev, void* out_dev, int V, int H, int W, int C, int S, void* stream) {
  return launch_adaptive_avgpool(dtype, in_dev, out_dev, V, H, W, C, S, (hipStream_t)stream);
}
int rgbm_build_volume(int dtype, const void* feat_dev, const float* P_views_dev, const float* depths_dev, float* homog_scratch,
                      void* vol_dev, int V, int B, int D, int H, int W, void* stream) {
  if (int rc = launch_homography(P_views_dev, homog_scratch, V, B, (hipStream_t)stream)) return rc;
  return launch_build_volume(dtype, feat_dev, homog_scratch, depths_dev, vol_dev, 0, V, V, B, D, H, W, (hipStream_t)stream);
}

}  // extern "C"

#include "prof.h"
extern "C" {
int rgbm_prof_rows(void) { return rgbm::kProfVariants; }
int rgbm_microbench_mfma_scratch_floats(int* n) {
  RGBM_REQUIRE(n, "microbench_mfma_scratch_floats arguments");
  return rgbm::microbench_mfma_scratch_floats(n);
}
int rgbm_microbench_mfma(float* scratch, int iters, int random_operands, double* flops, void* stream) {
  return rgbm::launch_microbench_mfma(scratch, iters, random_operands, flops, (hipStream_t)stream);
}
int rgbm_microbench_copy(const void* src, void* dst, size_t bytes, void* stream) {
  return rgbm::launch_microbench_copy(src, dst, bytes, (hipStream_t)stream);
}
int rgbm_prof_start(void) { return rgbm::prof_start(); }
int rgbm_prof_select(int row) {
  RGBM_REQUIRE(row >= -1 && row < rgbm::kProfVariants, "prof_select: row out of range");
  return rgbm::prof_select(row);
}
int rgbm_prof_stop(double* stats) {
  RGBM_REQUIRE(stats != nullptr, "prof_stop arguments");
  return rgbm::prof_stop(stats, rgbm::kProfVariants);
}
}

// Layer-level entry for the halo-tiled 3-D conv (tests): layer 0..6 = conv0..conv6, 7..9 = conv7/9/11 (transposed).
extern "C" int rgbm_conv3d_tile(int layer, int dtype, const void* in_dev, int N, int D, int H, int W, const float* w_host,
                                const float* bn_scale_host, const float* bn_shift_host, const void* res_dev, void* out_dev,
                                void* stream) {
  static const int cin[10] = {32, 8, 16, 16, 32, 32, 64, 64, 32, 16};
  static const int cout[10] = {8, 16, 16, 32, 32, 64, 64, 32, 16, 8};
  static const int stride[10] = {1, 2, 1, 2, 1, 2, 1, 1, 1, 1};
  RGBM_REQUIRE(layer >= 0 && layer < 10 && in_dev && w_host && bn_scale_host && bn_shift_host && out_dev, "conv3d_tile arguments");
  const bool tr = layer >= 7;
  const int coutp = cout[layer] < 16 ? 16 : cout[layer];
  std::vector<float> packed;
  conv3d_tile_pack(w_host, bn_scale_host, cin[layer], cout[layer], coutp, tr, dtype, packed);
  void* wdev = nullptr; float* bdev = nullptr;
  if (upload_packed(packed, dtype, &wdev)) return -2;
  std::vector<float> bpad(coutp, 0.f);
  for (int o = 0; o < cout[layer]; ++o) bpad[o] = bn_shift_host[o];
  if (upload_f32(bpad.data(), bpad.size(), &bdev)) return -2;
  Conv3dTileDesc d;
  memset(&d, 0, sizeof(d));
  d.in = in_dev; d.wgt = wdev; d.out = out_dev; d.bias = bdev; d.res = res_dev;
  d.N = N; d.Di = D; d.Hi = H; d.Wi = W;
  if (tr) { d.Do = 2 * D; d.Ho = 2 * H; d.Wo = 2 * W; d.Dq = D; d.Hq = H; d.Wq = W; }
  else {
    const int s = stride[layer];
    d.Do = (D + 2 - 3) / s + 1; d.Ho = (H + 2 - 3) / s + 1; d.Wo = (W + 2 - 3) / s + 1;
    d.Dq = d.Do; d.Hq = d.Ho; d.Wq = d.Wo;
  }
  d.Cout = cout[layer]; d.relu = 1; d.prof_variant = -1;
  int rc = launch_conv3d_tile(layer, dtype, d, (hipStream_t)stream);
  if (!rc) { hipError_t e = hipStreamSynchronize((hipStream_t)stream); if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = -2; } }
  (void)hipFree(wdev); (void)hipFree(bdev);
  return rc;
}

// Kernel-level entry for the depth-sweeping conv0 + fused plane sweep (tests): bf16 features [V][H][W][32] -> [V][D][H][W][8].
static int conv0_sweep_entry(int dtype, int feat_f16, const void* feat_dev, const float* P_views_dev, const float* depths_dev,
                             float* homog_scratch, const float* w_host, const float* bn_scale_host, const float* bn_shift_host,
                             void* out_dev, int V, int B, int D, int H, int W, void* stream) {
  RGBM_REQUIRE(feat_dev && P_views_dev && depths_dev && homog_scratch && w_host && bn_scale_host && bn_shift_host && out_dev,
               "conv0_sweep arguments");
  RGBM_REQUIRE(dtype == BF16 || dtype == F16 || dtype == BF16X3, "conv0_sweep: 16-bit storage types or bf16x3 (fp32 features in, split-pair c0 out)");
  if (int rc = launch_homography(P_views_dev, homog_scratch, V, B, (hipStream_t)stream)) return rc;
  std::vector<float> packed;
  conv0_sweep_pack(w_host, bn_scale_host, packed);
  void* wdev = nullptr; float* bdev = nullptr;
  if (dtype == BF16X3) { if (conv0_sweep_x3_upload(packed, &wdev)) return -2; }
  else if (upload_packed(packed, feat_f16 ? (int)F16 : dtype, &wdev)) return -2;
  std::vector<float> bpad(16, 0.f);
  for (int o = 0; o < 8; ++o) bpad[o] = bn_shift_host[o];
  if (upload_f32(bpad.data(), bpad.size(), &bdev)) return -2;
  Conv3dTileDesc d;
  memset(&d, 0, sizeof(d));
  d.wgt = wdev; d.out = out_dev; d.bias = bdev;
  d.N = V; d.Di = D; d.Hi = H; d.Wi = W; d.Do = D; d.Ho = H; d.Wo = W; d.Dq = D; d.Hq = H; d.Wq = W;
  d.Cout = 8; d.relu = 1; d.prof_variant = -1;
  d.feat = feat_dev; d.homog = homog_scratch; d.depths = depths_dev; d.v0 = 0; d.V = V; d.B = B;
  d.feat_f16 = feat_f16;
  int rc = dtype == BF16X3 ? launch_conv0_sweep_x3(d, (hipStream_t)stream) : launch_conv0_sweep(d, dtype, (hipStream_t)stream);
  if (!rc) { hipError_t e = hipStreamSynchronize((hipStream_t)stream); if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = -2; } }
  (void)hipFree(wdev); (void)hipFree(bdev);
  return rc;
}

extern "C" int rgbm_conv0_sweep_dt(int dtype, const void* feat_dev, const float* P_views_dev, const float* depths_dev,
                                   float* homog_scratch, const float* w_host, const float* bn_scale_host, const float* bn_shift_host,
                                   void* out_dev, int V, int B, int D, int H, int W, void* stream) {
  return conv0_sweep_entry(dtype, 0, feat_dev, P_views_dev, depths_dev, homog_scratch, w_host, bn_scale_host, bn_shift_host, out_dev, V, B, D,
                           H, W, stream);
}

extern "C" int rgbm_conv0_sweep_f16feat(const void* feat_f16_dev, const float* P_views_dev, const float* depths_dev,
                                        float* homog_scratch, const float* w_host, const float* bn_scale_host,
                                        const float* bn_shift_host, void* out_bf16_dev, int V, int B, int D, int H, int W, void* stream) {
  return conv0_sweep_entry(BF16, 1, feat_f16_dev, P_views_dev, depths_dev, homog_scratch, w_host, bn_scale_host, bn_shift_host, out_bf16_dev,
                           V, B, D, H, W, stream);
}

extern "C" int rgbm_conv0_sweep(const void* feat_dev, const float* P_views_dev, const float* depths_dev, float* homog_scratch,
                                const float* w_host, const float* bn_scale_host, const float* bn_shift_host, void* out_dev,
                                int V, int B, int D, int H, int W, void* stream) {
  return rgbm_conv0_sweep_dt(BF16, feat_dev, P_views_dev, depths_dev, homog_scratch, w_host, bn_scale_host, bn_shift_host, out_dev,
                             V, B, D, H, W, stream);
}

// Batched device-side prepare_model_input (interface_v5.py:58-170); see include/rgbm.h.
extern "C" int rgbm_prepare_inputs(const float* rgb_dev, const uint8_t* mask_dev, const double* K_dev, int N, int H, int W, int S,
                                   int P, uint32_t seed, float* img_out, int32_t* choose_out, float* pts2d_out, double* Kcrop_out,
                                   int32_t* window_out, int32_t* valid_out, uint8_t* scratch, void* stream) {
  return launch_prepare_inputs(rgb_dev, mask_dev, K_dev, nullptr, N, H, W, S, P, seed, img_out, choose_out, pts2d_out, Kcrop_out,
                               window_out, valid_out, scratch, (hipStream_t)stream);
}

extern "C" int rgbm_prepare_inputs_indexed(const float* rgb_dev, const uint8_t* mask_dev, const double* K_dev,
                                           const int32_t* frame_map_dev, int N, int H, int W, int S, int P, uint32_t seed,
                                           float* img_out, int32_t* choose_out, float* pts2d_out, double* Kcrop_out,
                                           int32_t* window_out, int32_t* valid_out, uint8_t* scratch, void* stream) {
  RGBM_REQUIRE(frame_map_dev, "prepare_inputs_indexed frame_map");
  return launch_prepare_inputs(rgb_dev, mask_dev, K_dev, frame_map_dev, N, H, W, S, P, seed, img_out, choose_out, pts2d_out,
                               Kcrop_out, window_out, valid_out, scratch, (hipStream_t)stream);
}

extern "C" int rgbm_prepare_inputs_ex(const float* rgb_dev, const uint8_t* mask_dev, const double* K_dev, const int32_t* frame_map_dev,
                                      int frame0, int N, int H, int W, int S, int P, uint32_t seed, float* img_out, int32_t* choose_out,
                                      float* pts2d_out, double* Kcrop_out, int32_t* window_out, int32_t* valid_out, uint8_t* scratch,
                                      void* stream) {
  RGBM_REQUIRE(frame0 >= 0, "prepare_inputs_ex frame0");
  return launch_prepare_inputs(rgb_dev, mask_dev, K_dev, frame_map_dev, N, H, W, S, P, seed, img_out, choose_out, pts2d_out,
                               Kcrop_out, window_out, valid_out, scratch, (hipStream_t)stream, frame0);
}

extern "C" int rgbm_adapose_postprocess_pnp(int B, int P, uint32_t seed, const float* nocs1, const float* pts2d1, const float* nocs2,
                                            const float* pts2d2, const double* K, const double* E1, const double* E2, double* bbox_out,
                                            double* srt_out, int32_t* info_out, int32_t* valid_out, void* stream) {
  return launch_pnp_ransac(nocs1, pts2d1, nocs2, pts2d2, K, E1, E2, bbox_out, srt_out, info_out, valid_out, B, P, seed, (hipStream_t)stream);
}

extern "C" int rgbm_projection(const double* Kcrop_dev, const double* E_dev, float* P_dev, int N, void* stream) {
  return launch_projection(Kcrop_dev, E_dev, P_dev, N, (hipStream_t)stream);
}

extern "C" int rgbm_mask_extent(const uint8_t* mask_dev, int N, int H, int W, int32_t* ext_out, int32_t* count_out, void* stream) {
  return launch_mask_extent(mask_dev, N, H, W, ext_out, count_out, (hipStream_t)stream);
}

// ---- controller step (control.hip) and synthetic camera (synth_env.hip) ---------------------------------------------
static_assert(sizeof(rgbm_control_reward_args) == sizeof(rgbm::ControlRewardArgs), "control reward args ABI mismatch");
static_assert(sizeof(rgbm_synth_scene) == sizeof(rgbm::SynthScene), "synth scene ABI mismatch");
extern "C" int rgbm_lookat_quat(const double* dir_dev, int N, int batch_zero, double* quat_dev, void* stream) {
  return rgbm::launch_lookat_quat(dir_dev, quat_dev, N, batch_zero, (hipStream_t)stream);
}
extern "C" int rgbm_control_action_to_pose(const float* action_dev, int lda, const double* pose_mid, const double* pose_min,
                                           const double* pose_max, int N, double* pose_dev, void* stream) {
  return rgbm::launch_control_action(action_dev, lda, pose_mid, pose_min, pose_max, pose_dev, N, (hipStream_t)stream);
}
extern "C" int rgbm_control_reward(const rgbm_control_reward_args* args, void* stream) {
  RGBM_REQUIRE(args, "control_reward args");
  return rgbm::launch_control_reward(*reinterpret_cast<const rgbm::ControlRewardArgs*>(args), (hipStream_t)stream);
}
extern "C" int rgbm_control_grasp_frame(const double* est_dev, int N, double* center_dev, double* direction_dev, void* stream) {
  return rgbm::launch_control_grasp_frame(est_dev, center_dev, direction_dev, N, (hipStream_t)stream);
}
extern "C" int rgbm_synth_camera(const rgbm_synth_scene* scene, double* K_dev, double* E_dev, double* rays_dev, void* stream) {
  RGBM_REQUIRE(scene, "synth_camera scene");
  return rgbm::launch_synth_camera(*reinterpret_cast<const rgbm::SynthScene*>(scene), K_dev, E_dev, rays_dev, (hipStream_t)stream);
}
extern "C" int rgbm_synth_render(const rgbm_synth_scene* scene, const double* rays_dev, float* color_dev, uint8_t* mask_dev,
                                 void* stream) {
  RGBM_REQUIRE(scene, "synth_render scene");
  return rgbm::launch_synth_render(*reinterpret_cast<const rgbm::SynthScene*>(scene), rays_dev, color_dev, mask_dev,
                                   (hipStream_t)stream);
}

extern "C" int rgbm_debug_flags(int flags) { rgbm::g_debug_flags = flags; ++rgbm::g_tuning_version; return 0; }

extern "C" int rgbm_set_tuning(const char* key, long long value) {
  RGBM_REQUIRE(key != nullptr, "set_tuning arguments");
  const std::string k = key;
  if (k == "ws_min_rows") { RGBM_REQUIRE(value >= 0, "ws_min_rows"); rgbm::g_ws_min_rows = value; }
  else if (k == "gemm_kernel") { RGBM_REQUIRE(value >= 0 && value <= 2, "gemm_kernel"); rgbm::g_gemm_kernel = (int)value; }
  else { set_error("unknown tuning key " + k); return -1; }
  ++rgbm::g_tuning_version;
  return 0;
}

// ---- PPO policy -------------------------------------------------------------------------------------------------
static_assert(sizeof(rgbm_policy_layout) == sizeof(rgbm::PolicyLayout), "policy layout ABI mismatch");
static_assert(sizeof(rgbm::PolicyOptState) == 40 || sizeof(rgbm::PolicyOptState) == 48, "opt state size");
extern "C" {
int rgbm_policy_forward(const float* params, const rgbm_policy_layout* L, int n, int mode, const float* obs, const float* noise,
                        float* actions, float* logp, float* value, float* mu, void* stream) {
  RGBM_REQUIRE(params && L && obs && mu, "policy_forward arguments");
  RGBM_REQUIRE(mode == 1 || (logp && value && actions), "policy_forward outputs");
  RGBM_REQUIRE(mode != 0 || noise, "policy_forward: act needs noise");
  return launch_policy_forward(params, *reinterpret_cast<const PolicyLayout*>(L), n, mode, obs, noise, actions, logp, value, mu,
                               (hipStream_t)stream);
}
int rgbm_ppo_partial_floats(const rgbm_policy_layout* L, int n, size_t* count) {
  RGBM_REQUIRE(L && count && n > 0, "ppo_partial_floats arguments");
  *count = (size_t)policy_partial_floats(*reinterpret_cast<const PolicyLayout*>(L), n);
  return 0;
}
int rgbm_ppo_minibatch_fwd_bwd(const float* params, const rgbm_policy_layout* L, int n, const float* obs, const float* actions,
                               const float* old_logp, const float* adv, const float* returns, const float* old_values,
                               const float* old_mu, const float* old_log_std, float clip, float vcoef, float ecoef,
                               float* partial_scratch, float* grads_flat, void* stream) {
  RGBM_REQUIRE(params && L && obs && actions && old_logp && adv && returns && old_values && old_mu && old_log_std &&
               partial_scratch && grads_flat, "ppo_minibatch arguments");
  return launch_ppo_minibatch(params, *reinterpret_cast<const PolicyLayout*>(L), n, obs, actions, old_logp, adv, returns,
                              old_values, old_mu, old_log_std, clip, vcoef, ecoef, partial_scratch, grads_flat, (hipStream_t)stream);
}
int rgbm_ppo_clip_adam(float* params, const float* grads_flat, float* exp_avg, float* exp_avg_sq, void* opt_state,
                       const rgbm_policy_layout* L, float inv_world, float max_norm, float desired_kl, float lr_min,
                       float lr_max, int adaptive, void* stream) {
  RGBM_REQUIRE(params && grads_flat && exp_avg && exp_avg_sq && opt_state && L, "ppo_clip_adam arguments");
  return launch_ppo_adam(params, grads_flat, exp_avg, exp_avg_sq, reinterpret_cast<PolicyOptState*>(opt_state), L->total,
                         inv_world, max_norm, desired_kl, lr_min, lr_max, adaptive, (hipStream_t)stream);
}
}
