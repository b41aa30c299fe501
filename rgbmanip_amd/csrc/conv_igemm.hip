// Generic implicit-GEMM convolution for gfx950 (CDNA4): 1-D/2-D/3-D, stride, dilation,
// sub-pixel (transposed-conv) output mapping, fused bias / residual / activation epilogue.
//
// Replaces the cuDNN conv calls behind the reference's nn.Conv2d / nn.Conv3d /
// nn.ConvTranspose3d / nn.Conv1d(k=1) / nn.Linear layers
// (/root/reference/models/pose_estimator/AdaPose/lib/pspnet.py:11-30,97-107,
//  lib/network_v5.py:8-28,217-291,317-376).
//
// GEMM view (per workgroup):   D[ch][pix] = sum_k W[ch][k] * X[k][pix]
//   A operand = weights  [BCH rows ][BK]  (K contiguous; pre-packed, BN folded, zero padded)
//   B operand = gathered [BPIX rows][BK]  input patch (NDHWC, k = tap*Cin + c)
// so every lane ends up with 4 consecutive output channels of one pixel (MFMA 16x16 C layout:
// col = lane&15 -> pixel, row = (lane>>4)*4+r -> channel) and stores them as one 8/16-byte write.
//
// 256 threads = 4 wave64; wave tile = min(BCH,64) channels x 64 pixels; K tile = 128 bytes per row
// (64 bf16 / 32 f32), register-staged global->LDS double buffering, one barrier per K tile.
// LDS rows are 128 B with a 16-B-chunk XOR swizzle (chunk ^= (row>>1)&7) so that the 16 rows a
// ds_read_b128 lane group touches land on 16 distinct 16-B bank slots.
// bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate); f32: v_mfma_f32_16x16x4_f32 (exact fp32).
#include <type_traits>
#include "common.h"
#include "prof.h"

namespace rgbm {

extern int g_debug_flags;
int launch_conv_glds(const ConvDesc& d, int dtype, hipStream_t s);


int conv_ch_tile(int Cout) {
  if (Cout <= 16) return 16;
  if (Cout <= 32) return 32;
  if (Cout <= 64) return 64;
  return 128;
}
int conv_bk(int dtype) { return 8 * dtype_chunk(dtype); }      // 128 bytes per row


int launch_conv(const ConvDesc& d, int dtype, hipStream_t s) {
  RGBM_REQUIRE(d.M > 0 && d.M < (1ll << 31), "conv M out of range");
  RGBM_REQUIRE(d.Cout % 4 == 0 && d.ldo % 4 == 0, "conv Cout/ldo must be multiples of 4");
  RGBM_REQUIRE(d.KT > 0 && d.Kpad == d.KT * conv_bk(dtype), "conv K padding mismatch");
  const int E = dtype_chunk(dtype);
  RGBM_REQUIRE(d.Cin % E == 0, "conv Cin must be a multiple of the 16-byte chunk");
  if (d.lcin >= 0) {
    RGBM_REQUIRE((1 << d.lcin) == d.Cin, "conv lcin mismatch");
  } else {
    RGBM_REQUIRE(d.ntaps == 1, "linear-K mode needs a single tap");
  }
  return launch_conv_glds(d, dtype, s);      // the LDS-DMA kernels (conv_igemm_glds.hip)
}

}  // namespace rgbm
