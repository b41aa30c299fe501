// Achievable-peak probes for bench.py (gfx950): SURVEY.md section 8(d) asks for the peaks the box actually reaches next to the
// datasheet ones — "measure achievable peaks on the box ... and use those as denominators too".  Two probes, timed by the caller
// with events on the launch stream:
//   * a bare v_mfma_f32_16x16x32_bf16 stream in the register shape of the implicit GEMM's multiply waves (8 waves per CU, 16
//     accumulators, 4 + 4 operand fragments, 32 MFMAs per iteration, no memory access), on constant operands (few bits toggle) or
//     on pseudo-random ones (the part's power limit lowers the clock: the rate a real GEMM can get);
//   * a grid-stride 16-byte copy (one read + one write stream).
#include "common.h"
#include "kernels.h"

namespace rgbm {

namespace {

__device__ __forceinline__ uint4 mb_fill(unsigned i, int random) {
  if (!random) return make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);      // every element 1.0
  unsigned h = i * 2654435761u + 12345u;
  uint4 v;
  unsigned* p = &v.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
    p[k] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);      // two bf16 of magnitude [0.5, 2), random sign and mantissa
  }
  return v;
}

__global__ __launch_bounds__(512) void mfma_stream_kernel(float* __restrict__ out, int iters, int random) {
  const int tid = threadIdx.x;
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = mb_fill((unsigned)(tid * 8 + i), random); b[i] = mb_fill((unsigned)(tid * 8 + 4 + i), random); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                                    acc[i * 4 + j], 0, 0, 0);
    // keep the accumulators bounded without leaving the matrix pipe idle for long: one scale every 64 iterations
    if ((it & 63) == 63) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] *= 0x1p-20f;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

__global__ __launch_bounds__(256) void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

}  // namespace

// scratch: n_cu * 512 floats (rgbm_microbench_mfma_scratch_floats); *flops = what this launch executes
int launch_microbench_mfma(float* scratch, int iters, int random_operands, double* flops, hipStream_t s) {
  int n_cu = 0;
  if (int rc = persistent_grid_cus(&n_cu)) return rc;
  RGBM_REQUIRE(scratch && iters > 0, "microbench_mfma arguments");
  hipLaunchKernelGGL(mfma_stream_kernel, dim3((unsigned)n_cu), dim3(512), 0, s, scratch, iters, random_operands ? 1 : 0);
  RGBM_CHECK_HIP(hipGetLastError());
  if (flops) *flops = (double)n_cu * 8.0 * (double)iters * 32.0 * (16.0 * 16.0 * 32.0 * 2.0);
  return 0;
}

int microbench_mfma_scratch_floats(int* n) {
  int n_cu = 0;
  if (int rc = persistent_grid_cus(&n_cu)) return rc;
  *n = n_cu * 512;
  return 0;
}

int launch_microbench_copy(const void* src, void* dst, size_t bytes, hipStream_t s) {
  RGBM_REQUIRE(src && dst && bytes >= 16 && (bytes & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "microbench_copy: 16-byte aligned buffers");
  int n_cu = 0;
  if (int rc = persistent_grid_cus(&n_cu)) return rc;
  hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)(n_cu * 16)), dim3(256), 0, s, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), bytes / 16);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
