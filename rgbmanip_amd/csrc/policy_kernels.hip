// PPO policy kernels (gfx950): fused actor-critic forward, fused PPO loss + backward, gradient reduction,
// gradient-clip + adaptive-LR + Adam — the arithmetic of
//   /root/reference/algo/ppo/ppo/module.py:73-107   (ActorCritic.act / act_inference / evaluate)
//   /root/reference/algo/ppo/ppo/ppo.py:472-528     (one minibatch of PPO.update)
// for the reference's MLPs (obs -> h0 -> h1 -> h2 -> out, ELU hidden activations, 36 985 parameters for the shipped cfg).
// The problem is launch/latency bound, so the design minimises launches and host syncs: one launch for
// forward+loss+backward of a minibatch (64 rows per workgroup, activations in LDS, weight rows as wave-uniform scalar
// loads), one deterministic cross-block reduction, one single-block optimiser step that also applies the KL-adaptive
// learning-rate rule on the device (the reference syncs the host twice per minibatch for that, ppo.py:486-495,527-528).
// The Gaussian is the reference's quirky one: scale_tril = diag(exp(log_std)^2), i.e. std = exp(2*log_std).
#include "common.h"
#include "kernels.h"

namespace rgbm {

constexpr int PK_ROWS = 64;      // rows per workgroup
constexpr int PK_MAXW = 128;     // max layer width
constexpr int PK_LD = PK_MAXW + 1;
constexpr float LOG_2PI = 1.8378770664093453f;

__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : expm1f(x); }

// one dense layer for the block's 64 rows: out[r][o] = act(b[o] + sum_i W[o][i] * in[r][i]); lane = row, wave = column group
__device__ __forceinline__ void dense_fwd(const float* __restrict__ W, const float* __restrict__ b, const float* in, int ldi,
                                          float* out, int ldo, int I, int O, bool act) {
  const int r = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int o = g; o < O; o += 4) {
    const float* wr = W + (long long)o * I;      // wave-uniform address -> scalar loads
    float acc = b[o];
    for (int i = 0; i < I; ++i) acc = fmaf(wr[i], in[r * ldi + i], acc);
    out[r * ldo + o] = act ? elu(acc) : acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward only: mode 0 = act (sample with supplied N(0,1) noise), 1 = act_inference (mean), 2 = evaluate(actions)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void policy_forward_kernel(const float* __restrict__ P, PolicyLayout L, int n, int mode,
                                                             const float* __restrict__ obs, const float* __restrict__ noise,
                                                             float* __restrict__ actions, float* __restrict__ logp,
                                                             float* __restrict__ value, float* __restrict__ mu_out) {
  __shared__ float bufA[PK_ROWS * PK_LD], bufB[PK_ROWS * PK_LD], a0[PK_ROWS * PK_LD];
  const int r = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * PK_ROWS + r;
  const bool live = row < n;
  for (int i = g; i < L.dims[0]; i += 4) a0[r * PK_LD + i] = live ? obs[row * L.dims[0] + i] : 0.f;
  __syncthreads();
  for (int net = 0; net < (mode == 1 ? 1 : 2); ++net) {
    const float* in = a0;
    float* outb = bufA;
    for (int l = 0; l < 4; ++l) {
      const int I = L.dims[l], O = (l == 3) ? (net == 0 ? L.dims[4] : 1) : L.dims[l + 1];
      dense_fwd(P + L.w[net][l], P + L.b[net][l], in, PK_LD, outb, PK_LD, I, O, l < 3);
      __syncthreads();
      in = outb;
      outb = (outb == bufA) ? bufB : bufA;
    }
    // `in` now holds the net's output
    if (net == 0) {
      const int A = L.dims[4];
      if (live && g == 0) {
        float lp = -0.5f * A * LOG_2PI;
        for (int k = 0; k < A; ++k) {
          const float m = in[r * PK_LD + k], ls = P[L.log_std + k];
          mu_out[row * A + k] = m;
          if (mode == 1) continue;
          float a;
          if (mode == 0) { a = m + expf(2.f * ls) * noise[row * A + k]; actions[row * A + k] = a; }
          else a = actions[row * A + k];
          const float d = a - m;
          lp += -(d * d) / (2.f * expf(4.f * ls)) - 2.f * ls;
        }
        if (mode != 1) logp[row] = lp;
      }
      __syncthreads();
    } else if (live && g == 0) {
      value[row] = in[r * PK_LD];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward + PPO loss + backward for one minibatch; partial gradients per workgroup
// ---------------------------------------------------------------------------------------------------------
// dW[o][i] += sum_r delta[r][o] * a[r][i]   (256 threads stride over the O*I elements; rows from LDS)
__device__ __forceinline__ void dense_wgrad(const float* delta, int ldd, const float* a, int lda, float* gW, float* gb, int I,
                                            int O, int nrows) {
  for (int e = threadIdx.x; e < O * I; e += 256) {
    const int o = e / I, i = e - o * I;
    float acc = 0.f;
    for (int r = 0; r < nrows; ++r) acc = fmaf(delta[r * ldd + o], a[r * lda + i], acc);
    gW[e] = acc;
  }
  for (int o = threadIdx.x; o < O; o += 256) {
    float acc = 0.f;
    for (int r = 0; r < nrows; ++r) acc += delta[r * ldd + o];
    gb[o] = acc;
  }
}
// dprev[r][i] = (sum_o W[o][i] * delta[r][o]) * elu'(a_prev)   with elu'(z) expressed through a = elu(z): a>0 ? 1 : a+1
__device__ __forceinline__ void dense_dgrad(const float* __restrict__ W, const float* delta, int ldd, const float* aprev, int lda,
                                            float* dprev, int I, int O) {
  const int r = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int i = g; i < I; i += 4) {
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(W[(long long)o * I + i], delta[r * ldd + o], acc);
    const float a = aprev[r * lda + i];
    dprev[r * ldd + i] = acc * (a > 0.f ? 1.f : a + 1.f);
  }
}

__global__ __launch_bounds__(256) void ppo_loss_grad_kernel(const float* __restrict__ P, PolicyLayout L, int n,
                                                            const float* __restrict__ obs, const float* __restrict__ actions,
                                                            const float* __restrict__ old_logp, const float* __restrict__ adv,
                                                            const float* __restrict__ returns, const float* __restrict__ old_values,
                                                            const float* __restrict__ old_mu, const float* __restrict__ old_sigma,
                                                            float clip, float vcoef, float ecoef, float* __restrict__ partial,
                                                            int pstride) {
  extern __shared__ float sm[];
  // activations a0 (obs), a1, a2, a3 of the current net, each [64][dims[l]+1]; two delta buffers [64][LDD]
  float* act_[4];
  int lda[4];
  int off = 0, wmax = L.dims[4];
  for (int l = 0; l < 4; ++l) { act_[l] = sm + off; lda[l] = L.dims[l] + 1; off += PK_ROWS * lda[l]; if (l > 0 && L.dims[l] > wmax) wmax = L.dims[l]; }
  const int LDD = wmax + 1;
  float* dA = sm + off;                  // delta of the layer being processed
  float* dB = dA + PK_ROWS * LDD;
  float* red = dB + PK_ROWS * LDD;       // [64][20]: row statistics (3) + log_std gradient contributions (<=16)
  const int r = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long row0 = (long long)blockIdx.x * PK_ROWS;
  const long long row = row0 + r;
  const int nrows = (int)((n - row0) < PK_ROWS ? (n - row0) : PK_ROWS);
  const bool live = r < nrows;
  const int A = L.dims[4];
  float* gout = partial + (long long)blockIdx.x * pstride;
  const float invn = 1.0f / (float)n;

  for (int i = g; i < L.dims[0]; i += 4) act_[0][r * lda[0] + i] = live ? obs[row * L.dims[0] + i] : 0.f;
  __syncthreads();

  for (int net = 0; net < 2; ++net) {
    // ---- forward, keeping every activation ----
    for (int l = 0; l < 3; ++l) {
      dense_fwd(P + L.w[net][l], P + L.b[net][l], act_[l], lda[l], act_[l + 1], lda[l + 1], L.dims[l], L.dims[l + 1], true);
      __syncthreads();
    }
    const int O = net == 0 ? A : 1;
    dense_fwd(P + L.w[net][3], P + L.b[net][3], act_[3], lda[3], dB, LDD, L.dims[3], O, false);   // dB temporarily holds the output
    __syncthreads();
    // ---- loss gradient w.r.t. the net output -> dA ----
    if (g == 0) {
      if (net == 0) {
        float lp = -0.5f * A * LOG_2PI, kl = 0.f;
        for (int k = 0; k < A; ++k) {
          const float m = dB[r * LDD + k], ls = P[L.log_std + k];
          const float a = live ? actions[row * A + k] : m;
          const float d = a - m;
          lp += -(d * d) / (2.f * expf(4.f * ls)) - 2.f * ls;
          if (live) {
            const float os = old_sigma[row * A + k], om = old_mu[row * A + k];
            const float eo = expf(os), en = expf(ls);
            kl += ls - os + (eo * eo + (om - m) * (om - m)) / (2.f * en * en) - 0.5f;     // ppo.py:482-483
          }
        }
        float dlp = 0.f, surr = 0.f;
        if (live) {
          const float ratio = expf(lp - old_logp[row]);
          const float ad = adv[row];
          const float s1 = -ad * ratio;
          const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
          const float s2 = -ad * rc;
          surr = fmaxf(s1, s2);
          // d max(s1,s2)/d logp: s1 branch -> -ad*ratio; clipped branch contributes only inside the clip range (then s1==s2)
          const bool inside = ratio > 1.f - clip && ratio < 1.f + clip;
          dlp = (s1 > s2 || inside) ? -ad * ratio : ((s1 == s2) ? 0.5f * -ad * ratio : 0.f);
        }
        for (int k = 0; k < A; ++k) {
          const float m = dB[r * LDD + k], ls = P[L.log_std + k];
          const float a = live ? actions[row * A + k] : m;
          const float d = a - m, iv = expf(-4.f * ls);
          dA[r * LDD + k] = live ? dlp * (d * iv) * invn : 0.f;                              // dL/dmu
          red[r * 20 + 4 + k] = live ? (dlp * (2.f * d * d * iv - 2.f) * invn) : 0.f;          // dL/dlog_std via logp
        }
        red[r * 20 + 0] = surr;
        red[r * 20 + 2] = kl;
      } else {
        float vl = 0.f, dv = 0.f;
        if (live) {
          const float v = dB[r * LDD], tv = old_values[row], rt = returns[row];
          const float diff = v - tv;
          const float vc = tv + fminf(fmaxf(diff, -clip), clip);
          const float l1 = (v - rt) * (v - rt), l2 = (vc - rt) * (vc - rt);
          vl = fmaxf(l1, l2);
          const bool inside = diff > -clip && diff < clip;
          if (l1 > l2) dv = 2.f * (v - rt);
          else if (l1 < l2) dv = inside ? 2.f * (vc - rt) : 0.f;
          else dv = inside ? 2.f * (v - rt) : (v - rt);      // tie: torch.max splits the gradient evenly
          dv *= vcoef * invn;
        }
        dA[r * LDD] = dv;
        red[r * 20 + 1] = vl;
      }
    }
    __syncthreads();
    // ---- backward through the four layers ----
    float* dcur = dA;
    float* dnext = dB;
    for (int l = 3; l >= 0; --l) {
      const int I = L.dims[l], Ol = (l == 3) ? O : L.dims[l + 1];
      dense_wgrad(dcur, LDD, act_[l], lda[l], gout + L.w[net][l], gout + L.b[net][l], I, Ol, PK_ROWS);
      if (l > 0) dense_dgrad(P + L.w[net][l], dcur, LDD, act_[l], lda[l], dnext, I, Ol);
      __syncthreads();
      float* t = dcur; dcur = dnext; dnext = t;
    }
    if (net == 0) {
      // log_std gradient (+ entropy term: entropy = const + 2*sum(log_std) for every row -> -ecoef * 2)
      if (threadIdx.x < A) {
        float acc = 0.f;
        for (int rr = 0; rr < PK_ROWS; ++rr) acc += red[rr * 20 + 4 + threadIdx.x];
        gout[L.log_std + threadIdx.x] = acc - ecoef * 2.f * (float)nrows * invn;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    float acc = 0.f;
    for (int rr = 0; rr < PK_ROWS; ++rr) acc += red[rr * 20 + threadIdx.x];
    gout[L.total + threadIdx.x] = acc;
  }
  if (threadIdx.x == 3) gout[L.total + 3] = (float)nrows;
}

// grads[e] = sum over workgroups (fixed order -> deterministic); e in [0, total+4)
__global__ void ppo_reduce_kernel(const float* __restrict__ partial, int nblk, int pstride, int count, float* __restrict__ grads) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  float acc = 0.f;
  for (int b = 0; b < nblk; ++b) acc += partial[(long long)b * pstride + e];
  grads[e] = acc;
}

// single-workgroup optimiser step: average over ranks, clip by global norm, KL-adaptive LR, Adam (torch defaults)
__global__ __launch_bounds__(1024) void ppo_adam_kernel(float* __restrict__ P, const float* __restrict__ grads, float* __restrict__ m,
                                                        float* __restrict__ v, PolicyOptState* __restrict__ st, int total,
                                                        float inv_world, float max_norm, float desired_kl, float lr_min,
                                                        float lr_max, int adaptive) {
  __shared__ double red[1024];
  __shared__ float s_coef, s_lr, s_bc1, s_bc2s;
  double acc = 0.0;
  for (int i = threadIdx.x; i < total; i += 1024) { const double gval = (double)grads[i] * inv_world; acc += gval * gval; }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    float coef = max_norm / (norm + 1e-6f);
    s_coef = coef > 1.f ? 1.f : coef;
    const float rows = grads[total + 3];
    const float kl_mean = grads[total + 2] / rows;
    float lr = st->lr;
    if (adaptive) {                                   // ppo.py:486-495 (asymmetric clamps kept as shipped)
      if (kl_mean > desired_kl * 2.0f) lr = fmaxf(lr_min, lr / 1.5f);
      else if (kl_mean < desired_kl / 2.0f && kl_mean > 0.0f) lr = fminf(lr_max, lr * 1.5f);
    }
    st->lr = lr;
    st->t += 1;
    st->sum_surr += (double)(grads[total + 0] / rows);
    st->sum_vloss += (double)(grads[total + 1] / rows);
    st->last_kl = kl_mean;
    st->last_norm = norm;
    st->n_updates += 1;
    const double bc1 = 1.0 - pow(0.9, (double)st->t), bc2 = 1.0 - pow(0.999, (double)st->t);
    s_lr = lr; s_bc1 = (float)bc1; s_bc2s = (float)sqrt(bc2);
  }
  __syncthreads();
  const float coef = s_coef, lr = s_lr, bc1 = s_bc1, bc2s = s_bc2s;
  for (int i = threadIdx.x; i < total; i += 1024) {
    const float gval = grads[i] * inv_world * coef;
    const float mi = 0.9f * m[i] + 0.1f * gval;
    const float vi = 0.999f * v[i] + 0.001f * gval * gval;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + 1e-8f;
    P[i] -= (lr / bc1) * (mi / denom);
  }
}

static int check_layout(const PolicyLayout& L) {
  for (int l = 0; l < 5; ++l) RGBM_REQUIRE(L.dims[l] > 0 && L.dims[l] <= PK_MAXW, "policy layer width must be in 1..128");
  RGBM_REQUIRE(L.dims[4] <= 16, "policy action dim must be <= 16");
  return 0;
}

int launch_policy_forward(const float* params, const PolicyLayout& L, int n, int mode, const float* obs, const float* noise,
                          float* actions, float* logp, float* value, float* mu, hipStream_t s) {
  if (int rc = check_layout(L)) return rc;
  RGBM_REQUIRE(n > 0 && mode >= 0 && mode <= 2, "policy_forward arguments");
  hipLaunchKernelGGL(policy_forward_kernel, dim3((n + PK_ROWS - 1) / PK_ROWS), dim3(256), 0, s, params, L, n, mode, obs, noise,
                     actions, logp, value, mu);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int policy_partial_floats(const PolicyLayout& L, int n) { return ((n + PK_ROWS - 1) / PK_ROWS) * (L.total + 4); }

int launch_ppo_minibatch(const float* params, const PolicyLayout& L, int n, const float* obs, const float* actions,
                         const float* old_logp, const float* adv, const float* returns, const float* old_values,
                         const float* old_mu, const float* old_sigma, float clip, float vcoef, float ecoef, float* partial,
                         float* grads, hipStream_t s) {
  if (int rc = check_layout(L)) return rc;
  RGBM_REQUIRE(n > 0, "ppo_minibatch rows");
  const int nblk = (n + PK_ROWS - 1) / PK_ROWS, pstride = L.total + 4;
  int wmax = L.dims[4], asum = 0;
  for (int l = 0; l < 4; ++l) { asum += L.dims[l] + 1; if (l > 0 && L.dims[l] > wmax) wmax = L.dims[l]; }
  const size_t lds = (size_t)(PK_ROWS * (asum + 2 * (wmax + 1)) + PK_ROWS * 20) * sizeof(float);
  RGBM_REQUIRE(lds <= 160 * 1024, "policy too wide for the LDS-resident backward pass");
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(ppo_loss_grad_kernel), (int)lds)) return rc;
  hipLaunchKernelGGL(ppo_loss_grad_kernel, dim3(nblk), dim3(256), lds, s, params, L, n, obs, actions, old_logp, adv, returns,
                     old_values, old_mu, old_sigma, clip, vcoef, ecoef, partial, pstride);
  RGBM_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(ppo_reduce_kernel, dim3((pstride + 255) / 256), dim3(256), 0, s, partial, nblk, pstride, pstride, grads);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_ppo_adam(float* params, const float* grads, float* m, float* v, PolicyOptState* st, int total, float inv_world,
                    float max_norm, float desired_kl, float lr_min, float lr_max, int adaptive, hipStream_t s) {
  hipLaunchKernelGGL(ppo_adam_kernel, dim3(1), dim3(1024), 0, s, params, grads, m, v, st, total, inv_world, max_norm, desired_kl,
                     lr_min, lr_max, adaptive);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
