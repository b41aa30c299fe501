// Per-sample BatchNorm3d for the cost-regularisation net (norm_mode = 1): the reference as shipped never leaves train mode
// (interface_v5.py:39-56) and always runs one pose per call, so every BatchNorm3d (network_v5.py:17-28, 246-252) normalises a
// sample with the biased mean / variance of its own D x H x W volume.  Three kernels per layer behind the un-normalised conv:
// deterministic two-stage sums in fp64 (block partials in a fixed order, no atomics), the per-(view, channel) scale / shift table,
// and normalise + ReLU (+ post-activation skip add) in place.  An opt-in parity mode, not a throughput path.
#include "common.h"
#include "kernels.h"

namespace rgbm {

namespace {

constexpr int BN_NB = 64;            // partial-sum blocks per view

template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, double* __restrict__ part, long long nvox, int C) {
  __shared__ double ls[256][4], lq[256][4];
  const int v = blockIdx.y, tid = threadIdx.x;
  const int cg = C >> 2;                                   // 4-channel groups per voxel; (BN_NB * 256) % cg == 0, so a thread always
  const long long nchunk = nvox * cg;                      // meets the same group
  const T* xv = x + (long long)v * nvox * C;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 + tid; i < nchunk; i += (long long)BN_NB * 256) {
    float f[4];
    load4(xv + i * 4, f);
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[e] += (double)f[e]; q[e] += (double)f[e] * (double)f[e]; }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { ls[tid][e] = s[e]; lq[tid][e] = q[e]; }
  __syncthreads();
  if (tid < C) {
    const int g = tid >> 2, e = tid & 3;
    double a = 0, b = 0;
    for (int j = g; j < 256; j += cg) { a += ls[j][e]; b += lq[j][e]; }      // fixed order
    double* o = part + (((long long)v * BN_NB + blockIdx.x) * C + tid) * 2;
    o[0] = a; o[1] = b;
  }
}

__global__ void bn_finalize_kernel(const double* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ ss, int V, int C, double inv_n, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * C) return;
  const int v = i / C, c = i - v * C;
  double a = 0, b = 0;
  for (int k = 0; k < BN_NB; ++k) {
    const double* p = part + (((long long)v * BN_NB + k) * C + c) * 2;
    a += p[0]; b += p[1];
  }
  const double mean = a * inv_n;
  double var = b * inv_n - mean * mean;                    // biased variance: what BatchNorm normalises with in training mode
  if (var < 0) var = 0;
  const float scale = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
  ss[2 * i] = scale;
  ss[2 * i + 1] = beta[c] - (float)mean * scale;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(T* __restrict__ y, const float* __restrict__ ss, const T* __restrict__ res,
                                                         long long nvox, int C, int V, int relu) {
  const int cg = C >> 2;
  const long long per_view = nvox * cg, total = per_view * V;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long v = i / per_view;
    const int g = (int)(i % cg);
    float f[4], r[4] = {0.f, 0.f, 0.f, 0.f};
    load4(y + i * 4, f);
    if (res) load4(res + i * 4, r);
    const float* t = ss + ((long long)v * C + g * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float o = f[e] * t[2 * e] + t[2 * e + 1];
      if (relu) o = o < 0.f ? 0.f : o;                     // NaN propagates, like torch.relu
      f[e] = o + r[e];                                     // skip adds are post-ReLU (network_v5.py:287-289)
    }
    store4(y + i * 4, f);
  }
}

template <typename T>
int run_t(void* y, const void* res, const float* gamma, const float* beta, double* part, float* ss, int V, long long nvox, int C,
          int relu, hipStream_t s) {
  hipLaunchKernelGGL(bn_partial_kernel<T>, dim3(BN_NB, V), dim3(256), 0, s, (const T*)y, part, nvox, C);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((V * C + 127) / 128), dim3(128), 0, s, part, gamma, beta, ss, V, C, 1.0 / (double)nvox, 1e-5f);
  const long long total = nvox * (C >> 2) * V;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(bn_apply_kernel<T>, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, (T*)y, ss, (const T*)res,
                     nvox, C, V, relu);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

size_t bn_scratch_bytes(int V) { return (size_t)V * BN_NB * 64 * 2 * sizeof(double) + (size_t)V * 64 * 2 * sizeof(float) + 256; }

// y [V][nvox][C] (conv output, no bias / BN / activation) -> relu(batchnorm_per_view(y)) + res, in place.  scratch: bn_scratch_bytes(V)
int launch_bn_per_sample(int dtype, void* y, const void* res, const float* gamma, const float* beta, void* scratch, int V,
                         long long nvox, int C, int relu, hipStream_t s) {
  RGBM_REQUIRE(C % 4 == 0 && C <= 64 && (BN_NB * 256) % (C / 4) == 0 && V > 0 && nvox > 0, "per-sample BatchNorm geometry");
  double* part = reinterpret_cast<double*>(scratch);
  float* ss = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + (size_t)V * BN_NB * 64 * 2 * sizeof(double));
  if (dtype == BF16) return run_t<unsigned short>(y, res, gamma, beta, part, ss, V, nvox, C, relu, s);
  if (dtype == F16) return run_t<f16_t>(y, res, gamma, beta, part, ss, V, nvox, C, relu, s);
  if (dtype == BF16X3) return run_t<bx3_t>(y, res, gamma, beta, part, ss, V, nvox, C, relu, s);
  return run_t<float>(y, res, gamma, beta, part, ss, V, nvox, C, relu, s);
}

}  // namespace rgbm
