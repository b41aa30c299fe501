// PPO rollout-storage kernels (gfx950): GAE backward recurrence and global advantage normalisation.
// Replaces the 16-step Python loop of tiny tensor ops in
//   /root/reference/algo/ppo/ppo/storage.py:50-64 (RolloutStorage.compute_returns).
// Layout [T][N] (t major), one thread per env, the recurrence lives in registers, loads/stores are
// coalesced across envs.  HBM-bound: 4 B*(3 reads + 2 writes) + 1 B per (t, env).
#include "common.h"
#include "kernels.h"

// The recurrence must round exactly like the reference's separate fp32 tensor ops: no a*b+c -> fma fusion
// (HIP's __fmul_rn/__fadd_rn are plain operators that the default -ffp-contract=fast would still fuse).
#pragma clang fp contract(off)

namespace rgbm {

__global__ __launch_bounds__(256) void gae_kernel(int T, int N, const float* __restrict__ rewards,
                                                  const unsigned char* __restrict__ dones, const float* __restrict__ values,
                                                  const float* __restrict__ last_values, float gamma, float lam,
                                                  float* __restrict__ returns, float* __restrict__ adv,
                                                  double* __restrict__ partial) {
  __shared__ double rs[256], rq[256];
  const int n = blockIdx.x * 256 + threadIdx.x;
  double s = 0.0, q = 0.0;
  if (n < N) {
    float a = 0.f;
    float nv = last_values[n];
    for (int t = T - 1; t >= 0; --t) {
      const long long i = (long long)t * N + n;
      const float v = values[i];
      const float m = 1.0f - (float)dones[i];
      // delta = r + m*gamma*V' - V ; A = delta + m*gamma*lam*A   (same association as storage.py:57-59)
      const float delta = __fsub_rn(__fadd_rn(rewards[i], __fmul_rn(__fmul_rn(m, gamma), nv)), v);
      a = __fadd_rn(delta, __fmul_rn(__fmul_rn(__fmul_rn(m, gamma), lam), a));
      const float ret = __fadd_rn(a, v);
      returns[i] = ret;
      const float ad = __fsub_rn(ret, v);      // advantages = returns - values (storage.py:63)
      adv[i] = ad;
      s += (double)ad;
      q += (double)ad * (double)ad;
      nv = v;
    }
  }
  rs[threadIdx.x] = s; rq[threadIdx.x] = q;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) { rs[threadIdx.x] += rs[threadIdx.x + st]; rq[threadIdx.x] += rq[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[blockIdx.x * 2] = rs[0]; partial[blockIdx.x * 2 + 1] = rq[0]; }
}

__global__ void gae_sum_partials_kernel(const double* __restrict__ partial, int nblk, double* __restrict__ sums) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < nblk; ++i) { s += partial[i * 2]; q += partial[i * 2 + 1]; }
    sums[0] = s; sums[1] = q;
  }
}

// sums = {sum(adv), sum(adv^2)} (device, fp64, possibly all-reduced over ranks), followed by scratch for
// the per-block partials.  Caller provides sums with room for 2 + 2*ceil(N/256) doubles.
int launch_gae(int T, int N, const float* rewards, const unsigned char* dones, const float* values, const float* last_values,
               float gamma, float lam, float* returns, float* adv, double* sums, hipStream_t s) {
  RGBM_REQUIRE(T > 0 && N > 0, "gae shape");
  const int nblk = (N + 255) / 256;
  hipLaunchKernelGGL(gae_kernel, dim3(nblk), dim3(256), 0, s, T, N, rewards, dones, values, last_values, gamma, lam, returns,
                     adv, sums + 2);
  RGBM_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(gae_sum_partials_kernel, dim3(1), dim3(64), 0, s, sums + 2, nblk, sums);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// adv <- (adv - mean) / (std_unbiased + 1e-8) with mean/std from the (global) sums  (storage.py:64)
__global__ void adv_normalise_kernel(long long n, float* __restrict__ adv, const double* __restrict__ sums, double count) {
  const double mean = sums[0] / count;
  double var = (sums[1] - count * mean * mean) / (count - 1.0);
  if (var < 0.0) var = 0.0;
  const float mf = (float)mean;
  const float denom = (float)sqrt(var) + 1e-8f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    adv[i] = (adv[i] - mf) / denom;
}

int launch_adv_normalise(long long n_local, float* adv, const double* sums, double count_total, hipStream_t s) {
  unsigned g = (unsigned)((n_local + 255) / 256);
  if (g > 2048) g = 2048;
  if (g == 0) g = 1;
  hipLaunchKernelGGL(adv_normalise_kernel, dim3(g), dim3(256), 0, s, n_local, adv, sums, count_total);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
