#pragma once
#include <hip/hip_runtime.h>
namespace rgbm {
// rows: 0..7 conv_igemm_glds_kernel dtype*4 + {0: BCH16, 1: BCH32, 2: BCH64, 3: BCH128}; 8/9 conv3d_tile_kernel f32/bf16
// (conv1..conv11); 10/11 conv3d_tile_kernel conv0 + fused warp (f32/bf16); 12/13 conv_igemm_ws_kernel (and its non-uniform-tap fallback conv_igemm_v3_kernel) f32/bf16;
// 14 conv0_sweep_kernel (bf16); 15 conv_igemm_ws64_kernel (bf16); 16..25 conv3d_tile_kernel bf16, one row per layer (conv0..conv6, conv7, conv9, conv11);
// 26..29 the bf16x3 kernels, 30 conv_igemm_w256_kernel, 31 / 32 conv_igemm_ws_kernel<bf16, wide> / <bf16, wide, row halo>, 33 conv_igemm_ws_kernel<bx3_t, wide>, 34 / 35 / 36 the 64-channel x 256-pixel four-multiply-wave shape of conv_igemm_ws_kernel for bx3_t / 16-bit / f32 tensors,
// 37 / 38 upconv_combine_kernel on 16-bit / 4-byte storage, 39 upconv_final_kernel.  Row 9 stays empty (the bf16 3-D layers are listed one by one); row 8 aggregates the f32 3-D layers.
constexpr int kProfVariants = 42;      // == RGBM_PROF_ROWS (include/rgbm.h)
bool prof_enabled();
void prof_begin_launch(hipStream_t s, int variant, double flops, double bytes);
void prof_end_launch(hipStream_t s);
int prof_start();
int prof_select(int variant);      // -1: every row (default); >= 0: only that row's launches get events (fewer events in a timed region)
int prof_stop(double* stats, int n_variants);
}  // namespace rgbm
