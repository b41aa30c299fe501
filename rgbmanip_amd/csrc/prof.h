#pragma once
#include <hip/hip_runtime.h>
namespace rgbm {
// variants: conv_igemm_kernel dtype*4 + {0: BCH16, 1: BCH32, 2: BCH64, 3: BCH128}; 8/9: conv3d_tile_kernel f32/bf16
constexpr int kProfVariants = 12;   // 10/11: conv3d_tile conv0 (f32/bf16) on its own
bool prof_enabled();
void prof_begin_launch(hipStream_t s, int variant, double flops, double bytes);
void prof_end_launch(hipStream_t s);
int prof_start();
int prof_stop(double* stats, int n_variants);
}  // namespace rgbm
