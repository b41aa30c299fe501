// Host-side layer objects: weight packing (BN folding, NDHWC/K-major repack, bf16 conversion) and
// ConvDesc construction for the generic implicit-GEMM kernel.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace rgbm {

struct HostTensor {
  const float* data = nullptr;   // fp32 host pointer (int64 tensors such as num_batches_tracked are skipped)
  std::vector<long long> shape;
  long long numel() const { long long n = 1; for (auto s : shape) n *= s; return n; }
};
typedef std::map<std::string, HostTensor> StateDict;

struct ConvGeom {
  int Cin = 0, Cout = 0;         // logical channels
  int KD = 1, KH = 1, KW = 1;
  int sd = 1, sh = 1, sw = 1;
  int pd = 0, ph = 0, pw = 0;
  int dild = 1, dilh = 1, dilw = 1;
  bool transposed = false;       // ConvTranspose3d k3 s2 p1 op1 (sub-pixel decomposition)
  int act = ACT_NONE;
  float slope = 0.f;
};

// One packed weight set (a transposed conv owns 8, one per output parity class).
struct PackedConv {
  void* w = nullptr;             // [Cout_pad][Kpad] in the layer dtype
  int KD = 1, KH = 1, KW = 1, ntaps = 1, Kpad = 0, KT = 0;
};

struct ConvLayer {
  ConvGeom g;
  int dtype = F32;
  int Cin_pad = 0;               // physical input channels (>= Cin, multiple of the 16-byte chunk)
  int Cout_pad = 0;              // physical output channels written (multiple of 4)
  float* bias = nullptr;         // [Cout_pad] fp32 or null
  std::vector<PackedConv> packs; // 1, or 8 for transposed
  std::vector<void*> owned;      // device allocations to free
  mutable bool out_plain_f32 = false;    // (set per run by the owner) bf16x3 layers only: the output tensor is plain fp32 (for what gathers from it), see ConvDesc::out_f32

  // weights: [Cout][Cin][KD][KH][KW] (or [Cin][Cout][3][3][3] when transposed), optional per-Cout scale/shift
  // (folded BN) and bias.  Cin_pad: physical channel count of the input tensor.
  int init(int dtype, const ConvGeom& g, const float* w, const float* bias, const float* bn_scale, const float* bn_shift,
           int Cin_pad, int Cout_pad);
  void destroy();

  // Run on input [N][Di][Hi][Wi][Cin_pad] -> output [N][Do][Ho][Wo][ldo] (+ch offset via out pointer).
  // res: optional residual with the output's layout.  bias_override/bias_stride: per-sample bias.
  int run(const void* in, void* out, int N, int Di, int Hi, int Wi, int ldo, const void* res, int res_mode,
          const float* bias_override, int bias_stride, hipStream_t s) const;
  void out_dims(int Di, int Hi, int Wi, int& Do, int& Ho, int& Wo) const;
  // this conv followed by `next` (a 1x1, stride 1): one fused launch writing only out2 when the kernel selection allows it
  // (*fused = true; `mid` is then NOT written), otherwise the two layers one after the other through `mid`.
  int run_then_1x1(const ConvLayer& next, const void* in, void* mid, int ldmid, void* out2, int ldo2, int N, int Di, int Hi, int Wi,
                   bool allow_fuse, bool* fused, hipStream_t s) const;
  int build_desc(ConvDesc& d, const void* in, void* out, int N, int Di, int Hi, int Wi, int ldo, const void* res, int res_mode,
                 const float* bias_override, int bias_stride, int cls) const;
};

// PSPUpsample (x2 bilinear align_corners -> conv3x3 pad 1 + bias -> activation; pspnet.py:100-107) as a 1x1 GEMM at the low
// resolution with the nine taps stacked on the output channels, followed by the tap-combining kernel (upconv.hip)
struct UpConvLayer {
  ConvLayer gemm;                // 1x1: Cin -> 9*Cout rows in (tap, channel) order, no bias, no activation
  ConvLayer gemm2;               // round 6: the stacked rows behind the last multiple of 256 (up_2: 576 = 512 + 64), see init()
  int split = 0;                 // rows of `gemm` when the product runs as two launches (0: one launch)
  float* bias = nullptr;         // [Cout] fp32 or null
  int Cout = 0, act = ACT_NONE;
  float slope = 0.f;
  // w: [Cout][Cin][3][3] (nn.Conv2d layout)
  int init(int dtype, int Cin, int Cout, const float* w, const float* bias_h, int act, float slope);
  void destroy();
  size_t scratch_elems(int V, int h, int w) const { return (size_t)V * h * w * 9 * Cout; }
  // in [V][h][w][Cin] -> z scratch [V][h][w][9*Cout] -> out [V][2h][2w][ldo]
  int run(const void* in, void* z, void* out, int V, int h, int w, int ldo, hipStream_t s) const;
};

// PSPNet tail: up_3 (x2 bilinear -> conv3x3 64 -> 64 + bias -> PReLU) and `final` (conv1x1 64 -> 32 + bias) in ONE kernel
// (upconv_final.hip): neither the up-sampled tensor, nor the tap-stacked z, nor up_3's output reach memory.  16-bit / split pairs.
struct UpConvFinal {
  void* wz = nullptr;            // [9 * 64][64] storage type, rows in (tap, channel) order
  void* wf = nullptr;            // [32][64] storage type
  void* wf_h = nullptr;          // bf16 nets: the same in f16 (out_kind 2: the f16 tail)
  float* bias = nullptr;         // [64] up_3 bias
  float* biasf = nullptr;        // [32] final bias
  float slope = 0.f;
  int dtype = 0;
  bool ready() const { return wz != nullptr; }
  bool f16_ready() const { return wf_h != nullptr; }      // `final`'s weights are representable in f16 (weights_fit_f16): the f16 tail exists
  // w3 [64][64][3][3], b3 [64], wfin [32][64], bfin [32] (nn.Conv2d layouts)
  int init(int dtype, const float* w3, const float* b3, float slope, const float* wfin, const float* bfin);
  void destroy();
  // in [V][h][w][64] -> out [V][2h][2w][32] (storage type; plain fp32: split-pair path only; f16: bf16 path only)
  int run(const void* in, void* out, int out_kind, int V, int h, int w, hipStream_t s) const;      // out_kind: launch_upconv_final's out_f32 (0 storage type, 1 plain fp32, 2 f16)
};

// device upload helpers
bool weights_fit_f16(const std::vector<float>& w);      // no value saturates in IEEE f16, no appreciable weight mass below its smallest normal
int upload_packed(const std::vector<float>& w, int dtype, void** dev);      // host fp32 -> device array in storage type dtype
int upload_f32(const float* host, size_t n, float** dev);

}  // namespace rgbm
