// The `direct_regression: False`, `use_depth: True` tail of AdaPoseEstimator_v5.predict on the GPU:
//   /root/reference/models/pose_estimator/AdaPose/interface_v5.py:322-339  back-projection of the predicted depth
//   /root/reference/models/pose_estimator/AdaPose/lib/align.py:10-41       estimateSimilarityUmeyama
//   /root/reference/models/pose_estimator/AdaPose/lib/align.py:44-104      estimateSimilarityTransform (RANSAC)
//   /root/reference/models/pose_estimator/AdaPose/interface_v5.py:348-374  bbox from (s, R, t), world frame, default bbox
// One workgroup per pose.  The reference runs 128 RANSAC iterations one after the other, each a 5-point Umeyama fit and a
// residual pass over all P points, stopping early by a confidence rule.  Here the 128 hypotheses are fitted by 128 threads at
// once, the P x 128 residual tests are spread over the waves (ballot + popcount), and thread 0 then replays the
// reference's sequential "strictly better ratio / early break" scan over the 128 inlier counts, so the hypothesis chosen is
// the one the sequential loop would have kept.  The final fit over the inliers uses block-wide fp64 sums.
// The 5-point samples come from a seeded hash (sample k of iteration i of pose b = mix32(seed, 128 b + i, k) mod P) where
// the reference uses the global np.random.randint — the same distribution, reproducible, restated in oracle/align_ref.py.
// A NaN covariance makes the reference raise; here the pose is marked invalid (default bbox).
#include "common.h"
#include "kernels.h"
#include "bbox_emit.h"

#pragma clang fp contract(off)

namespace rgbm {

namespace {

constexpr int AL_THREADS = 256;
constexpr int AL_MAXP = 1024;
constexpr int AL_ITERS = 128;

__device__ __forceinline__ unsigned al_mix32(unsigned seed, unsigned frame, unsigned idx) {      // = mix32 of prepare.hip
  unsigned h = seed ^ (frame * 0x9E3779B9u) ^ (idx * 0x85EBCA6Bu);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

__device__ inline double det3(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// A (row-major 3x3) = U diag(S) V^T, S descending, by one-sided Jacobi on the columns of A.
__device__ void svd3(const double* A, double* U, double* S, double* V) {
  double W[9], Vm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) W[i] = A[i];
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) { alpha += W[r * 3 + p] * W[r * 3 + p]; beta += W[r * 3 + q] * W[r * 3 + q]; gamma += W[r * 3 + p] * W[r * 3 + q]; }
        if (gamma == 0.0 || fabs(gamma) <= 1e-300) continue;
        off = fmax(off, fabs(gamma) / sqrt(alpha * beta));
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
        for (int r = 0; r < 3; ++r) {
          const double wp = W[r * 3 + p], wq = W[r * 3 + q];
          W[r * 3 + p] = c * wp - s * wq; W[r * 3 + q] = s * wp + c * wq;
          const double vp = Vm[r * 3 + p], vq = Vm[r * 3 + q];
          Vm[r * 3 + p] = c * vp - s * vq; Vm[r * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sv[3];
  for (int j = 0; j < 3; ++j) sv[j] = sqrt(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
  int ord[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (sv[ord[b]] > sv[ord[a]]) { const int tmp = ord[a]; ord[a] = ord[b]; ord[b] = tmp; }
  const double tiny = 1e-14 * fmax(sv[ord[0]], 1e-300);
  for (int j = 0; j < 3; ++j) {
    const int o = ord[j];
    S[j] = sv[o];
    for (int r = 0; r < 3; ++r) { V[r * 3 + j] = Vm[r * 3 + o]; U[r * 3 + j] = sv[o] > tiny ? W[r * 3 + o] / sv[o] : 0.0; }
  }
  // columns of U that belong to (numerically) zero singular values: complete to an orthonormal basis
  if (!(S[0] > tiny)) { U[0] = 1; U[3] = 0; U[6] = 0; }
  if (!(S[1] > tiny)) {
    const double ax = fabs(U[0]), ay = fabs(U[3]), az = fabs(U[6]);
    double e[3] = {0, 0, 0};
    e[(ax <= ay && ax <= az) ? 0 : (ay <= az ? 1 : 2)] = 1.0;
    const double d = e[0] * U[0] + e[1] * U[3] + e[2] * U[6];
    double v[3] = {e[0] - d * U[0], e[1] - d * U[3], e[2] - d * U[6]};
    const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    U[1] = v[0] / n; U[4] = v[1] / n; U[7] = v[2] / n;
  }
  if (!(S[2] > tiny)) {
    U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
  }
}

// Umeyama from sufficient statistics: n, centroids ms / mt, Cov = sum (t - mt)(s - ms)^T / n, varP = sum_axis var(source).
// -> scale, R (row-major), t.  Returns false for a NaN covariance.
__device__ bool umeyama_from_stats(const double* cov, const double* ms, const double* mt, double varP, double& scale, double* R, double* t) {
  for (int i = 0; i < 9; ++i) if (cov[i] != cov[i]) return false;
  double U[9], S[3], V[9];
  svd3(cov, U, S, V);
  double Vh[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Vh[i * 3 + j] = V[j * 3 + i];
  if (det3(U) * det3(Vh) < 0.0) { S[2] = -S[2]; U[2] = -U[2]; U[5] = -U[5]; U[8] = -U[8]; }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = U[i * 3] * Vh[j] + U[i * 3 + 1] * Vh[3 + j] + U[i * 3 + 2] * Vh[6 + j];
  scale = 1 / varP * ((S[0] + S[1]) + S[2]);
  for (int j = 0; j < 3; ++j) t[j] = mt[j] - ((ms[0] * (scale * R[j * 3]) + ms[1] * (scale * R[j * 3 + 1])) + ms[2] * (scale * R[j * 3 + 2]));
  return true;
}

__device__ double al_block_sum(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = AL_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

}  // namespace

__global__ __launch_bounds__(AL_THREADS) void umeyama_ransac_kernel(
    const float* __restrict__ nocs /*[B,P,3]*/, const float* __restrict__ depth /*[B,P]*/, const int* __restrict__ choose /*[B,P]*/,
    const double* __restrict__ Kc /*[B,9]*/, const double* __restrict__ E1 /*[B,16]*/, double* __restrict__ bbox /*[B,8,3]*/,
    double* __restrict__ srt /*[B,13]: s, R(9), t(3)*/, int* __restrict__ valid, int P, int img, unsigned seed) {
  __shared__ double sx[AL_MAXP], sy[AL_MAXP], sz[AL_MAXP], tx[AL_MAXP], ty[AL_MAXP], tz[AL_MAXP];
  __shared__ double hyp[AL_ITERS][13];          // s*R (9), t (3), threshold
  __shared__ int cnt[AL_ITERS];
  __shared__ double red[AL_THREADS];
  __shared__ int s_best, s_fail;
  __shared__ float hmax[3];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double fx = Kc[b * 9 + 0], cx = Kc[b * 9 + 2], fy = Kc[b * 9 + 4], cy = Kc[b * 9 + 5];
  if (t == 0) { s_best = -1; s_fail = 0; hmax[0] = hmax[1] = hmax[2] = 0.f; }
  __syncthreads();
  double ax = 0, ay = 0, az = 0;
  bool nan_in = false;
  for (int p = t; p < P; p += AL_THREADS) {
    const float* n = nocs + ((long long)b * P + p) * 3;
    sx[p] = n[0]; sy[p] = n[1]; sz[p] = n[2];
    const int ch = choose[(long long)b * P + p];
    const double z = (double)depth[(long long)b * P + p];
    tx[p] = ((double)(ch % img) - cx) * z / fx;
    ty[p] = ((double)(ch / img) - cy) * z / fy;
    tz[p] = z;
    ax += sx[p]; ay += sy[p]; az += sz[p];
    nan_in |= (n[0] != n[0]) || (n[1] != n[1]) || (n[2] != n[2]);
    atomicMax((int*)&hmax[0], __float_as_int(fabsf(n[0])));
    atomicMax((int*)&hmax[1], __float_as_int(fabsf(n[1])));
    atomicMax((int*)&hmax[2], __float_as_int(fabsf(n[2])));
  }
  // inlier threshold: a tenth of the source diameter (align.py:52-57)
  const double mx = al_block_sum(ax, red) / P, my = al_block_sum(ay, red) / P, mz = al_block_sum(az, red) / P;
  double far = 0.0;
  for (int p = t; p < P; p += AL_THREADS) {
    const double dx = sx[p] - mx, dy = sy[p] - my, dz = sz[p] - mz;
    far = fmax(far, sqrt((dx * dx + dy * dy) + dz * dz));
  }
  red[t] = far;
  __syncthreads();
  for (int s = AL_THREADS / 2; s > 0; s >>= 1) { if (t < s) red[t] = fmax(red[t], red[t + s]); __syncthreads(); }
  const double inlier_t = 2 * red[0] / 10.0;
  __syncthreads();

  // ---- 128 five-point hypotheses, one per thread (align.py:66-70) ----
  if (t < AL_ITERS) {
    int idx[5];
    double ms[3] = {0, 0, 0}, mt[3] = {0, 0, 0};
    for (int k = 0; k < 5; ++k) {
      idx[k] = (int)(al_mix32(seed, (unsigned)b * AL_ITERS + t, k) % (unsigned)P);
      ms[0] += sx[idx[k]]; ms[1] += sy[idx[k]]; ms[2] += sz[idx[k]];
      mt[0] += tx[idx[k]]; mt[1] += ty[idx[k]]; mt[2] += tz[idx[k]];
    }
    for (int j = 0; j < 3; ++j) { ms[j] /= 5; mt[j] /= 5; }
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, var = 0.0;
    for (int k = 0; k < 5; ++k) {
      const double cs[3] = {sx[idx[k]] - ms[0], sy[idx[k]] - ms[1], sz[idx[k]] - ms[2]};
      const double ct[3] = {tx[idx[k]] - mt[0], ty[idx[k]] - mt[1], tz[idx[k]] - mt[2]};
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cov[i * 3 + j] += ct[i] * cs[j];
      var += (cs[0] * cs[0] + cs[1] * cs[1]) + cs[2] * cs[2];
    }
    for (int i = 0; i < 9; ++i) cov[i] /= 5;
    var /= 5;
    double scale, R[9], tr[3];
    if (!umeyama_from_stats(cov, ms, mt, var, scale, R, tr)) { s_fail = 1; scale = 0; for (int i = 0; i < 9; ++i) R[i] = 0; tr[0] = tr[1] = tr[2] = 0; }
    for (int i = 0; i < 9; ++i) hyp[t][i] = scale * R[i];
    hyp[t][9] = tr[0]; hyp[t][10] = tr[1]; hyp[t][11] = tr[2];
    hyp[t][12] = scale * inlier_t;
  }
  __syncthreads();

  // ---- inlier counts of every hypothesis: wave w takes hypotheses w, w+4, ... (align.py:71-76) ----
  for (int h = wave; h < AL_ITERS; h += AL_THREADS / 64) {
    const double* m = hyp[h];
    int c = 0;
    for (int p = lane; p < P; p += 64) {
      const double rx = tx[p] - (((m[0] * sx[p] + m[1] * sy[p]) + m[2] * sz[p]) + m[9]);
      const double ry = ty[p] - (((m[3] * sx[p] + m[4] * sy[p]) + m[5] * sz[p]) + m[10]);
      const double rz = tz[p] - (((m[6] * sx[p] + m[7] * sy[p]) + m[8] * sz[p]) + m[11]);
      c += __popcll(__ballot(sqrt((rx * rx + ry * ry) + rz * rz) < m[12]));
    }
    if (lane == 0) cnt[h] = c;
  }
  __syncthreads();

  // ---- the reference's sequential scan: strictly better ratio wins, early break by confidence (align.py:77-87) ----
  if (t == 0) {
    double best = 0.0;
    int best_h = -1;
    for (int i = 0; i < AL_ITERS; ++i) {
      const double ratio = (double)cnt[i] / (double)P;
      if (ratio > best) { best = ratio; best_h = i; }
      const double b5 = (best * best) * (best * best) * best;
      if ((1 - pow(1 - b5, (double)i)) > 0.99) break;
    }
    s_best = best < 0.1 ? -1 : best_h;
  }
  __syncthreads();
  const int best_h = s_best;
  const bool fail = s_fail != 0;

  // ---- final fit over the inliers of the kept hypothesis (align.py:93-95) ----
  double n_in = 0, a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
  if (best_h >= 0) {
    const double* m = hyp[best_h];
    for (int p = t; p < P; p += AL_THREADS) {
      const double rx = tx[p] - (((m[0] * sx[p] + m[1] * sy[p]) + m[2] * sz[p]) + m[9]);
      const double ry = ty[p] - (((m[3] * sx[p] + m[4] * sy[p]) + m[5] * sz[p]) + m[10]);
      const double rz = tz[p] - (((m[6] * sx[p] + m[7] * sy[p]) + m[8] * sz[p]) + m[11]);
      if (sqrt((rx * rx + ry * ry) + rz * rz) < m[12]) {
        n_in += 1; a0 += sx[p]; a1 += sy[p]; a2 += sz[p]; b0 += tx[p]; b1 += ty[p]; b2 += tz[p];
      } else {
        sx[p] = __builtin_nan("");              // mark as outlier for the second pass (this thread owns p)
      }
    }
  }
  const double n = al_block_sum(n_in, red);
  const double ms[3] = {al_block_sum(a0, red) / n, al_block_sum(a1, red) / n, al_block_sum(a2, red) / n};
  const double mt[3] = {al_block_sum(b0, red) / n, al_block_sum(b1, red) / n, al_block_sum(b2, red) / n};
  double cv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, var = 0.0;
  if (best_h >= 0) {
    for (int p = t; p < P; p += AL_THREADS) {
      if (sx[p] != sx[p]) continue;
      const double cs[3] = {sx[p] - ms[0], sy[p] - ms[1], sz[p] - ms[2]};
      const double ct[3] = {tx[p] - mt[0], ty[p] - mt[1], tz[p] - mt[2]};
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cv[i * 3 + j] += ct[i] * cs[j];
      var += (cs[0] * cs[0] + cs[1] * cs[1]) + cs[2] * cs[2];
    }
  }
  double cov[9];
  for (int i = 0; i < 9; ++i) cov[i] = al_block_sum(cv[i], red) / n;
  var = al_block_sum(var, red) / n;
  const double any_nan = al_block_sum(nan_in ? 1.0 : 0.0, red);
  if (t == 0) {
    double scale = 0, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tr[3] = {0, 0, 0};
    bool ok = best_h >= 0 && !fail && any_nan == 0.0;
    if (ok) ok = umeyama_from_stats(cov, ms, mt, var, scale, R, tr);
    double* o = srt + (long long)b * 13;
    o[0] = ok ? scale : __builtin_nan("");
    for (int i = 0; i < 9; ++i) o[1 + i] = R[i];
    for (int i = 0; i < 3; ++i) o[10 + i] = tr[i];
    // sRT is a float32 matrix in the reference (interface_v5.py:357-361): R and t are rounded to float32 there
    double Rf[9];
    for (int i = 0; i < 9; ++i) Rf[i] = (double)(float)R[i];
    const float tf[3] = {(float)tr[0], (float)tr[1], (float)tr[2]};
    const double size[3] = {2.0 * (double)hmax[0] * scale, 2.0 * (double)hmax[1] * scale, 2.0 * (double)hmax[2] * scale};
    emit_bbox_world(b, Rf, tf, size, ok, E1, bbox, valid);
  }
}

int launch_umeyama_ransac(const float* nocs, const float* depth, const int* choose, const double* Kc, const double* E1,
                          double* bbox, double* srt, int* valid, int B, int P, int img, unsigned seed, hipStream_t s) {
  RGBM_REQUIRE(nocs && depth && choose && Kc && E1 && bbox && srt && valid, "umeyama_ransac arguments");
  RGBM_REQUIRE(B > 0 && P >= 5 && P <= AL_MAXP && img > 0, "umeyama_ransac needs 5 <= P <= 1024");
  hipLaunchKernelGGL(umeyama_ransac_kernel, dim3(B), dim3(AL_THREADS), 0, s, nocs, depth, choose, Kc, E1, bbox, srt, valid, P,
                     img, seed);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
