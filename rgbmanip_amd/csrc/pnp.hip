// The `direct_regression: False`, `use_depth: False` tail of AdaPoseEstimator_v5.predict on the GPU (the PnP branch):
//   /root/reference/models/pose_estimator/AdaPose/interface_v5.py:340-346   P = K E[:3], NOCS matches, estimatePnPRansac
//   /root/reference/models/pose_estimator/AdaPose/lib/utils.py:121-195      depth_estimation_from_nocs_matches (mutual NOCS nearest
//                                                                          neighbours, 0.01 gate, epipolar gate, cv2.triangulatePoints,
//                                                                          compute_scale of the left points)
//   /root/reference/models/pose_estimator/AdaPose/lib/utils.py:76-96        compute_scale (median of pair-distance ratios)
//   /root/reference/models/pose_estimator/AdaPose/lib/align.py:104-115      cv2.solvePnPRansac(EPNP, reprojectionError 3) ->
//                                                                          cv2.solvePnPRefineVVS -> cv2.Rodrigues
//   /root/reference/models/pose_estimator/AdaPose/interface_v5.py:348-374   bbox from (scale, R, t), world frame, default bbox
// The OpenCV routines are restated from their published algorithms (oracle/pnp_ref.py lists them; parity with cv2 itself is
// UNPINNED: OpenCV is not available in the build image).  One workgroup per pose, everything in fp64 except where the reference
// computes in float32 (NOCS distances, the skew matrix of the epipolar gate, nocs * scale, the pixels handed to solvePnP).
// RANSAC subsets come from the package's seeded hash (five distinct indices per iteration) where OpenCV uses its process-global
// MWC generator; the 100 five-point EPnP hypotheses are fitted by 100 threads at once, inlier counts by the whole block, and
// thread 0 replays OpenCV's sequential "better count -> shrink the iteration budget" scan over them.  Not a throughput path.
#include "common.h"
#include "kernels.h"
#include "bbox_emit.h"

#pragma clang fp contract(off)

namespace rgbm {

namespace {

constexpr int PN_T = 256;
constexpr int PN_P = 1024;
constexpr int PN_ITERS = 100;
constexpr int PN_MODEL = 5;

__device__ __forceinline__ unsigned pn_mix32(unsigned seed, unsigned frame, unsigned idx) {      // = mix32 of prepare.hip
  unsigned h = seed ^ (frame * 0x9E3779B9u) ^ (idx * 0x85EBCA6Bu);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

// ------------------------------------------------------------------------------------------------ small dense linear algebra
// symmetric n x n (row-major, n <= 12): cyclic Jacobi; on return ev[] ascending, V columns = eigenvectors in that order
__device__ void eig_sym(double* A, int n, double* ev, double* V) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j]; }
    if (!(off > 1e-60 * (diag + off))) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i) ev[i] = A[i * n + i];
  for (int i = 0; i < n - 1; ++i) {                         // selection sort, ascending, columns of V along
    int m = i;
    for (int j = i + 1; j < n; ++j) if (ev[j] < ev[m]) m = j;
    if (m != i) {
      const double te = ev[i]; ev[i] = ev[m]; ev[m] = te;
      for (int k = 0; k < n; ++k) { const double tv = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = tv; }
    }
  }
}

// least squares A x = b, A m x k (row-major, leading dimension k, m <= 6, k <= 6) by Householder QR; false if rank deficient
__device__ bool ls_solve(const double* A0, const double* b0, int m, int k, double* x) {
  double A[36], b[6];
  for (int i = 0; i < m * k; ++i) A[i] = A0[i];
  for (int i = 0; i < m; ++i) b[i] = b0[i];
  for (int c = 0; c < k; ++c) {
    double nrm = 0.0;
    for (int r = c; r < m; ++r) nrm += A[r * k + c] * A[r * k + c];
    nrm = sqrt(nrm);
    if (!(nrm > 0.0)) return false;
    const double alpha = A[c * k + c] > 0 ? -nrm : nrm;
    double v[6];
    for (int r = c; r < m; ++r) v[r] = A[r * k + c];
    v[c] -= alpha;
    double vn = 0.0;
    for (int r = c; r < m; ++r) vn += v[r] * v[r];
    if (vn > 0.0) {
      for (int j = c; j < k; ++j) {
        double d = 0.0;
        for (int r = c; r < m; ++r) d += v[r] * A[r * k + j];
        d = 2.0 * d / vn;
        for (int r = c; r < m; ++r) A[r * k + j] -= d * v[r];
      }
      double d = 0.0;
      for (int r = c; r < m; ++r) d += v[r] * b[r];
      d = 2.0 * d / vn;
      for (int r = c; r < m; ++r) b[r] -= d * v[r];
    }
  }
  for (int c = k - 1; c >= 0; --c) {
    double s = b[c];
    for (int j = c + 1; j < k; ++j) s -= A[c * k + j] * x[j];
    if (!(fabs(A[c * k + c]) > 0.0)) return false;
    x[c] = s / A[c * k + c];
  }
  return true;
}

__device__ inline double det3p(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
__device__ bool inv3(const double* m, double* o) {
  const double d = det3p(m);
  if (!(fabs(d) > 0.0)) return false;
  o[0] = (m[4] * m[8] - m[5] * m[7]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
  o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[2] * m[3] - m[0] * m[5]) / d;
  o[6] = (m[3] * m[7] - m[4] * m[6]) / d; o[7] = (m[1] * m[6] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[1] * m[3]) / d;
  return true;
}
// R = U V^T of the SVD of the 3x3 matrix M (closest rotation up to the determinant fix of the caller): polar factor through the
// eigen-decomposition of M^T M (M = U S V^T  ->  U = M V S^-1)
__device__ void polar_uv(const double* M, double* R) {
  double MtM[9], ev[3], V[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) MtM[i * 3 + j] = M[0 * 3 + i] * M[0 * 3 + j] + M[1 * 3 + i] * M[1 * 3 + j] + M[2 * 3 + i] * M[2 * 3 + j];
  eig_sym(MtM, 3, ev, V);                                  // ascending
  double U[9];
  for (int c = 0; c < 3; ++c) {
    const double s = sqrt(fmax(ev[c], 0.0));
    for (int r = 0; r < 3; ++r) {
      const double mv = M[r * 3 + 0] * V[0 * 3 + c] + M[r * 3 + 1] * V[1 * 3 + c] + M[r * 3 + 2] * V[2 * 3 + c];
      U[r * 3 + c] = s > 0.0 ? mv / s : 0.0;
    }
  }
  if (!(ev[0] > 1e-24 * fmax(ev[2], 1e-300))) {            // rank 2: complete U's first column (smallest singular value) by the cross product
    U[0] = U[4] * U[8] - U[7] * U[5]; U[3] = U[7] * U[2] - U[1] * U[8]; U[6] = U[1] * U[5] - U[4] * U[2];
  }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] = U[i * 3 + 0] * V[j * 3 + 0] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
}

// ------------------------------------------------------------------------------------------------ EPnP (one thread)
struct PnpData {                       // LDS-resident per pose
  const double* pwx; const double* pwy; const double* pwz;       // object points (nocs * scale, float32 values)
  const double* u; const double* v;                               // pixels (float32 values)
  double fu, fv, uc, vc;
};

// idx: point list (n >= 4).  R row-major, t.  Returns false when no candidate gives a finite pose.
__device__ bool epnp(const PnpData& D, const int* idx, int n, double* R, double* t) {
  double cws[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < n; ++i) { const int p = idx[i]; cws[0][0] += D.pwx[p]; cws[0][1] += D.pwy[p]; cws[0][2] += D.pwz[p]; }
  for (int k = 0; k < 3; ++k) cws[0][k] /= n;
  double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const int p = idx[i];
    const double d[3] = {D.pwx[p] - cws[0][0], D.pwy[p] - cws[0][1], D.pwz[p] - cws[0][2]};
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[a * 3 + b] += d[a] * d[b];
  }
  double dc[3], uc3[9];
  eig_sym(C, 3, dc, uc3);                                  // ascending; control axis i uses the i-th LARGEST
  for (int i = 1; i < 4; ++i) {
    const int col = 3 - i;
    const double k = sqrt(fmax(dc[col], 0.0) / n);
    for (int j = 0; j < 3; ++j) cws[i][j] = cws[0][j] + k * uc3[j * 3 + col];
  }
  double CC[9], CCi[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) CC[j * 3 + i] = cws[i + 1][j] - cws[0][j];      // columns = control axes
  if (!inv3(CC, CCi)) return false;
  auto alphas = [&](int p, double* a) {
    const double d[3] = {D.pwx[p] - cws[0][0], D.pwy[p] - cws[0][1], D.pwz[p] - cws[0][2]};
    for (int j = 0; j < 3; ++j) a[1 + j] = CCi[j * 3 + 0] * d[0] + CCi[j * 3 + 1] * d[1] + CCi[j * 3 + 2] * d[2];
    a[0] = 1.0 - a[1] - a[2] - a[3];
  };
  double MtM[144];
  for (int i = 0; i < 144; ++i) MtM[i] = 0.0;
  for (int i = 0; i < n; ++i) {
    const int p = idx[i];
    double a[4], r1[12], r2[12];
    alphas(p, a);
    for (int j = 0; j < 4; ++j) {
      r1[3 * j] = a[j] * D.fu; r1[3 * j + 1] = 0.0; r1[3 * j + 2] = a[j] * (D.uc - D.u[p]);
      r2[3 * j] = 0.0; r2[3 * j + 1] = a[j] * D.fv; r2[3 * j + 2] = a[j] * (D.vc - D.v[p]);
    }
    for (int x = 0; x < 12; ++x) for (int y = x; y < 12; ++y) MtM[x * 12 + y] += r1[x] * r1[y] + r2[x] * r2[y];
  }
  for (int x = 0; x < 12; ++x) for (int y = 0; y < x; ++y) MtM[x * 12 + y] = MtM[y * 12 + x];
  double ev[12], V[144];
  eig_sym(MtM, 12, ev, V);                                 // V[:, 0] = smallest eigenvalue = OpenCV's ut[11]
  const int pr[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
  double L[6][10], rho[6];
  for (int r = 0; r < 6; ++r) {
    double dv[4][3];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) dv[i][k] = V[(3 * pr[r][0] + k) * 12 + i] - V[(3 * pr[r][1] + k) * 12 + i];
    auto dot = [&](int a, int b) { return dv[a][0] * dv[b][0] + dv[a][1] * dv[b][1] + dv[a][2] * dv[b][2]; };
    L[r][0] = dot(0, 0); L[r][1] = 2 * dot(0, 1); L[r][2] = dot(1, 1); L[r][3] = 2 * dot(0, 2); L[r][4] = 2 * dot(1, 2);
    L[r][5] = dot(2, 2); L[r][6] = 2 * dot(0, 3); L[r][7] = 2 * dot(1, 3); L[r][8] = 2 * dot(2, 3); L[r][9] = dot(3, 3);
    rho[r] = 0.0;
    for (int k = 0; k < 3; ++k) { const double d = cws[pr[r][0]][k] - cws[pr[r][1]][k]; rho[r] += d * d; }
  }
  double best_err = 1e300;
  bool have = false;
  for (int cand = 0; cand < 3; ++cand) {
    double b[4] = {0, 0, 0, 0};
    bool ok = true;
    if (cand == 0) {
      const int cols[4] = {0, 1, 3, 6};
      double A[24], x[4];
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 4; ++c) A[r * 4 + c] = L[r][cols[c]];
      ok = ls_solve(A, rho, 6, 4, x);
      const double s = x[0] < 0 ? -1.0 : 1.0;
      b[0] = sqrt(s * x[0]); b[1] = s * x[1] / b[0]; b[2] = s * x[2] / b[0]; b[3] = s * x[3] / b[0];
    } else if (cand == 1) {
      double A[18], x[3];
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) A[r * 3 + c] = L[r][c];
      ok = ls_solve(A, rho, 6, 3, x);
      if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
      else { b[0] = sqrt(x[0]); b[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
      if (x[1] < 0) b[0] = -b[0];
    } else {
      double A[30], x[5];
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 5; ++c) A[r * 5 + c] = L[r][c];
      ok = ls_solve(A, rho, 6, 5, x);
      if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
      else { b[0] = sqrt(x[0]); b[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
      if (x[1] < 0) b[0] = -b[0];
      b[2] = x[3] / b[0];
    }
    for (int it = 0; it < 5 && ok; ++it) {                  // Gauss-Newton on the four betas
      double A[24], bb[6], x[4];
      for (int r = 0; r < 6; ++r) {
        const double* l = L[r];
        A[r * 4 + 0] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
        A[r * 4 + 1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
        A[r * 4 + 2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
        A[r * 4 + 3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
        bb[r] = rho[r] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2] +
                          l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
      }
      ok = ls_solve(A, bb, 6, 4, x);
      for (int k = 0; k < 4; ++k) b[k] += x[k];
    }
    if (!ok || !(isfinite(b[0]) && isfinite(b[1]) && isfinite(b[2]) && isfinite(b[3]))) continue;
    // control points in the camera frame, sign, absolute orientation
    double ccs[4][3];
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) {
      double s = 0.0;
      for (int i = 0; i < 4; ++i) s += b[i] * V[(3 * j + k) * 12 + i];
      ccs[j][k] = s;
    }
    auto pc_of = [&](int p, double* pc) {
      double a[4];
      alphas(p, a);
      for (int k = 0; k < 3; ++k) pc[k] = a[0] * ccs[0][k] + a[1] * ccs[1][k] + a[2] * ccs[2][k] + a[3] * ccs[3][k];
    };
    {
      double pc[3];
      pc_of(idx[0], pc);
      if (pc[2] < 0.0) for (int j = 0; j < 4; ++j) for (int k = 0; k < 3; ++k) ccs[j][k] = -ccs[j][k];
    }
    double pc0[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) { double pc[3]; pc_of(idx[i], pc); for (int k = 0; k < 3; ++k) pc0[k] += pc[k]; }
    for (int k = 0; k < 3; ++k) pc0[k] /= n;
    double ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
      const int p = idx[i];
      double pc[3];
      pc_of(p, pc);
      const double dw[3] = {D.pwx[p] - cws[0][0], D.pwy[p] - cws[0][1], D.pwz[p] - cws[0][2]};
      for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) ABt[a * 3 + c] += (pc[a] - pc0[a]) * dw[c];
    }
    double Rc[9], tc[3];
    polar_uv(ABt, Rc);
    if (det3p(Rc) < 0) { Rc[6] = -Rc[6]; Rc[7] = -Rc[7]; Rc[8] = -Rc[8]; }
    for (int k = 0; k < 3; ++k) tc[k] = pc0[k] - (Rc[k * 3] * cws[0][0] + Rc[k * 3 + 1] * cws[0][1] + Rc[k * 3 + 2] * cws[0][2]);
    double err = 0.0;
    for (int i = 0; i < n; ++i) {
      const int p = idx[i];
      const double X = Rc[0] * D.pwx[p] + Rc[1] * D.pwy[p] + Rc[2] * D.pwz[p] + tc[0];
      const double Y = Rc[3] * D.pwx[p] + Rc[4] * D.pwy[p] + Rc[5] * D.pwz[p] + tc[1];
      const double Z = Rc[6] * D.pwx[p] + Rc[7] * D.pwy[p] + Rc[8] * D.pwz[p] + tc[2];
      const double du = D.u[p] - (D.uc + D.fu * X / Z), dv = D.v[p] - (D.vc + D.fv * Y / Z);
      err += sqrt(du * du + dv * dv);
    }
    err /= n;
    if (isfinite(err) && err < best_err) {
      best_err = err; have = true;
      for (int i = 0; i < 9; ++i) R[i] = Rc[i];
      for (int i = 0; i < 3; ++i) t[i] = tc[i];
    }
  }
  return have;
}

__device__ double pn_block_sum(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = PN_T / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}
__device__ unsigned pn_block_count(unsigned v, unsigned* cnt) {
  if (threadIdx.x == 0) *cnt = 0u;
  __syncthreads();
  if (v) atomicAdd(cnt, v);
  __syncthreads();
  const unsigned r = *cnt;
  __syncthreads();
  return r;
}

}  // namespace

__global__ __launch_bounds__(PN_T) void pnp_ransac_kernel(
    const float* __restrict__ nocs1 /*[B,P,3]*/, const float* __restrict__ pts1 /*[B,P,2]*/, const float* __restrict__ nocs2,
    const float* __restrict__ pts2, const double* __restrict__ Kin /*[B,9]*/, const double* __restrict__ E1in /*[B,16]*/,
    const double* __restrict__ E2in, double* __restrict__ bbox /*[B,8,3]*/, double* __restrict__ srt /*[B,13]: scale, R(9), t(3)*/,
    int* __restrict__ info /*[B,4]: matches, ransac ok, inliers, hypotheses looked at*/, int* __restrict__ valid, int P, unsigned seed) {
  __shared__ float n1[PN_P][3], n2[PN_P][3];
  __shared__ double p1u[PN_P], p1v[PN_P], p2u[PN_P], p2v[PN_P];
  __shared__ double wx[PN_P], wy[PN_P], wz[PN_P];            // triangulated left points of the matches, later nocs1 * scale
  __shared__ int l2r[PN_P], r2l[PN_P], ml[PN_P];
  __shared__ double hyp[PN_ITERS][12];
  __shared__ int hcnt[PN_ITERS], hok[PN_ITERS];
  __shared__ double red[PN_T];
  __shared__ unsigned cnt_s;
  __shared__ int s_nm, s_best, s_used;
  __shared__ double s_F[9], s_scale, s_R[9], s_t[3];
  __shared__ float hmax[3];
  const int b = blockIdx.x, t = threadIdx.x;
  const double* K = Kin + (long long)b * 9;
  const double* E1 = E1in + (long long)b * 16;
  const double* E2 = E2in + (long long)b * 16;
  if (t == 0) { hmax[0] = hmax[1] = hmax[2] = 0.f; s_best = -1; s_used = 0; }
  __syncthreads();
  for (int p = t; p < P; p += PN_T) {
    for (int k = 0; k < 3; ++k) { n1[p][k] = nocs1[((long long)b * P + p) * 3 + k]; n2[p][k] = nocs2[((long long)b * P + p) * 3 + k]; }
    p1u[p] = (double)pts1[((long long)b * P + p) * 2]; p1v[p] = (double)pts1[((long long)b * P + p) * 2 + 1];
    p2u[p] = (double)pts2[((long long)b * P + p) * 2]; p2v[p] = (double)pts2[((long long)b * P + p) * 2 + 1];
    for (int k = 0; k < 3; ++k) atomicMax((int*)&hmax[k], __float_as_int(fabsf(n1[p][k])));
  }
  __syncthreads();
  // ---- mutual nearest neighbours in NOCS space (utils.py:124-137): float32 distances, first minimum like np.argmin ----
  for (int p = t; p < P; p += PN_T) {
    float best = INFINITY; int bi = 0;
    float best2 = INFINITY; int bi2 = 0;
    for (int q = 0; q < P; ++q) {
      const float dx = n1[p][0] - n2[q][0], dy = n1[p][1] - n2[q][1], dz = n1[p][2] - n2[q][2];
      const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
      if (d < best) { best = d; bi = q; }
      const float ex = n1[q][0] - n2[p][0], ey = n1[q][1] - n2[p][1], ez = n1[q][2] - n2[p][2];
      const float e = sqrtf((ex * ex + ey * ey) + ez * ez);
      if (e < best2) { best2 = e; bi2 = q; }
    }
    l2r[p] = bi; r2l[p] = bi2;
  }
  // ---- fundamental matrix of the epipolar gate (utils.py:146-158), thread 0 ----
  if (t == 0) {
    // rel = E1 inv(E2): inverse by Gauss-Jordan with partial pivoting
    double a[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = E2[i * 4 + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
      int piv = c; double bst = fabs(a[c][c]);
      for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > bst) { bst = fabs(a[r][c]); piv = r; }
      if (piv != c) for (int k = 0; k < 8; ++k) { const double tmp = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = tmp; }
      const double inv = 1.0 / a[c][c];
      for (int k = 0; k < 8; ++k) a[c][k] *= inv;
      for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k]; }
    }
    double rel[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += E1[i * 4 + k] * a[k][4 + j];
      rel[i * 4 + j] = s;
    }
    const double t1[3] = {rel[3], rel[7], rel[11]};
    double tx[9] = {0, (double)(float)(-t1[2]), (double)(float)t1[1], (double)(float)t1[2], 0, (double)(float)(-t1[0]),
                    (double)(float)(-t1[1]), (double)(float)t1[0], 0};      // float32 skew matrix, as in the reference
    double Ki[9];
    inv3(K, Ki);
    double A[9], Bm[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i * 3 + j] = Ki[0 * 3 + i] * tx[0 * 3 + j] + Ki[1 * 3 + i] * tx[1 * 3 + j] + Ki[2 * 3 + i] * tx[2 * 3 + j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Bm[i * 3 + j] = A[i * 3 + 0] * rel[0 * 4 + j] + A[i * 3 + 1] * rel[1 * 4 + j] + A[i * 3 + 2] * rel[2 * 4 + j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s_F[i * 3 + j] = Bm[i * 3 + 0] * Ki[0 * 3 + j] + Bm[i * 3 + 1] * Ki[1 * 3 + j] + Bm[i * 3 + 2] * Ki[2 * 3 + j];
  }
  __syncthreads();
  // ---- the three gates, compaction in left-index order (utils.py:134-172), thread 0 ----
  if (t == 0) {
    int nm = 0;
    for (int p = 0; p < P; ++p) {
      const int q = l2r[p];
      if (r2l[q] != p) continue;
      const float dx = n1[p][0] - n2[q][0], dy = n1[p][1] - n2[q][1], dz = n1[p][2] - n2[q][2];
      if (!(sqrtf((dx * dx + dy * dy) + dz * dz) < 0.01f)) continue;
      const double l[3] = {p1u[p], p1v[p], 1.0}, r[3] = {p2u[q], p2v[q], 1.0};
      double e = 0.0;
      for (int i = 0; i < 3; ++i) e += l[i] * ((s_F[i * 3] * r[0] + s_F[i * 3 + 1] * r[1]) + s_F[i * 3 + 2] * r[2]);
      if (!(fabs(e) < 1.0)) continue;
      ml[nm++] = p;
    }
    s_nm = nm;
  }
  __syncthreads();
  const int nm = s_nm;
  // ---- DLT triangulation of every match (cv2.triangulatePoints), then the left camera frame (utils.py:185-189) ----
  for (int m = t; m < nm; m += PN_T) {
    const int p = ml[m], q = l2r[p];
    double P1[12], P2[12];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) {
      P1[i * 4 + j] = (K[i * 3] * E1[0 * 4 + j] + K[i * 3 + 1] * E1[1 * 4 + j]) + K[i * 3 + 2] * E1[2 * 4 + j];
      P2[i * 4 + j] = (K[i * 3] * E2[0 * 4 + j] + K[i * 3 + 1] * E2[1 * 4 + j]) + K[i * 3 + 2] * E2[2 * 4 + j];
    }
    double A[16];
    for (int j = 0; j < 4; ++j) {
      A[0 * 4 + j] = p1u[p] * P1[8 + j] - P1[j];
      A[1 * 4 + j] = p1v[p] * P1[8 + j] - P1[4 + j];
      A[2 * 4 + j] = p2u[q] * P2[8 + j] - P2[j];
      A[3 * 4 + j] = p2v[q] * P2[8 + j] - P2[4 + j];
    }
    double AtA[16], ev[4], V[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += A[k * 4 + i] * A[k * 4 + j];
      AtA[i * 4 + j] = s;
    }
    eig_sym(AtA, 4, ev, V);
    const double X[4] = {V[0] / V[12], V[4] / V[12], V[8] / V[12], 1.0};
    wx[m] = ((E1[0] * X[0] + E1[1] * X[1]) + E1[2] * X[2]) + E1[3];
    wy[m] = ((E1[4] * X[0] + E1[5] * X[1]) + E1[6] * X[2]) + E1[7];
    wz[m] = ((E1[8] * X[0] + E1[9] * X[1]) + E1[10] * X[2]) + E1[11];
  }
  __syncthreads();
  // ---- compute_scale of the matched points (utils.py:76-96): exact median of the valid pair ratios by bisection on the bit pattern of
  //      the positive doubles (i < j pairs: the reference's full n x n list holds every ratio twice, same median) ----
  auto ratio_of = [&](int i, int j, double& r) -> bool {
    const int pi = ml[i], pj = ml[j];
    const float dx = n1[pi][0] - n1[pj][0], dy = n1[pi][1] - n1[pj][1], dz = n1[pi][2] - n1[pj][2];
    const float nd = sqrtf((dx * dx + dy * dy) + dz * dz);
    if (!(nd > 0.01f)) return false;
    const double ex = wx[i] - wx[j], ey = wy[i] - wy[j], ez = wz[i] - wz[j];
    const double rd = sqrt((ex * ex + ey * ey) + ez * ez);
    if (!(rd < 0.3)) return false;
    r = rd / (double)nd;
    return true;
  };
  const long long npairs = (long long)nm * (nm - 1) / 2;
  auto count_lt = [&](unsigned long long key, bool le) -> unsigned {       // #ratios < key (le: <= key), whole block
    unsigned c = 0;
    for (long long q = t; q < npairs; q += PN_T) {
      // pair index -> (i, j), i < j
      int i = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
      while ((long long)(i + 1) * (i + 2) / 2 <= q) ++i;
      while ((long long)i * (i + 1) / 2 > q) --i;
      const int j = (int)(q - (long long)i * (i + 1) / 2);
      double r;
      if (!ratio_of(i + 1, j, r)) continue;
      const unsigned long long k = (unsigned long long)__double_as_longlong(r);
      c += le ? (k <= key) : (k < key);
    }
    return pn_block_count(c, &cnt_s);
  };
  double scale = __builtin_nan("");
  {
    const unsigned total = count_lt(0x7ff0000000000000ull, true);          // all finite positive ratios (<= +inf)
    if (total > 0) {
      auto kth = [&](unsigned k) -> unsigned long long {                    // k-th smallest (0-based)
        unsigned long long res = 0ull;
        for (int bit = 62; bit >= 0; --bit) {
          const unsigned long long trial = res | (1ull << bit);
          if (count_lt(trial, false) <= k) res = trial;
        }
        return res;
      };
      const unsigned long long lo = kth((total - 1) / 2);
      unsigned long long hi = lo;
      if ((total & 1u) == 0u) hi = kth(total / 2);
      scale = (__longlong_as_double((long long)lo) + __longlong_as_double((long long)hi)) / 2.0;
    }
  }
  if (t == 0) s_scale = scale;
  __syncthreads();
  const bool have_scale = isfinite(scale);
  // ---- PnP inputs: nocs1 (float32) * float32(scale), pixels of view 1 (align.py:105, interface_v5.py:345) ----
  for (int p = t; p < P; p += PN_T) {
    const float sf = (float)scale;
    wx[p] = (double)(n1[p][0] * sf); wy[p] = (double)(n1[p][1] * sf); wz[p] = (double)(n1[p][2] * sf);
  }
  __syncthreads();
  PnpData D;
  D.pwx = wx; D.pwy = wy; D.pwz = wz; D.u = p1u; D.v = p1v; D.fu = K[0]; D.fv = K[4]; D.uc = K[2]; D.vc = K[5];
  // ---- 100 five-point EPnP hypotheses, one per thread ----
  if (t < PN_ITERS) {
    int idx[PN_MODEL], have = 0;
    for (unsigned k = 0; have < PN_MODEL; ++k) {
      const int c = (int)(pn_mix32(seed, (unsigned)b * 128u + (unsigned)t, k) % (unsigned)P);
      bool dup = false;
      for (int j = 0; j < have; ++j) dup |= idx[j] == c;
      if (!dup) idx[have++] = c;
    }
    double R[9], tv[3];
    const bool ok = have_scale && epnp(D, idx, PN_MODEL, R, tv);
    hok[t] = ok ? 1 : 0;
    for (int i = 0; i < 9; ++i) hyp[t][i] = ok ? R[i] : 0.0;
    for (int i = 0; i < 3; ++i) hyp[t][9 + i] = ok ? tv[i] : 0.0;
  }
  __syncthreads();
  auto sq_err = [&](const double* m, int p) -> double {
    const double X = ((m[0] * wx[p] + m[1] * wy[p]) + m[2] * wz[p]) + m[9];
    const double Y = ((m[3] * wx[p] + m[4] * wy[p]) + m[5] * wz[p]) + m[10];
    const double Z = ((m[6] * wx[p] + m[7] * wy[p]) + m[8] * wz[p]) + m[11];
    const double du = p1u[p] - (D.fu * X / Z + D.uc), dv = p1v[p] - (D.fv * Y / Z + D.vc);
    return du * du + dv * dv;
  };
  for (int h = 0; h < PN_ITERS; ++h) {
    unsigned c = 0;
    if (hok[h]) for (int p = t; p < P; p += PN_T) c += sq_err(hyp[h], p) <= 9.0;
    const unsigned n_in = pn_block_count(c, &cnt_s);
    if (t == 0) hcnt[h] = (int)n_in;
  }
  __syncthreads();
  // ---- OpenCV's sequential scan (ptsetreg.cpp): a strictly better count wins and shrinks the iteration budget ----
  if (t == 0) {
    int best = 0, best_h = -1, niters = PN_ITERS, it = 0;
    while (it < niters) {
      const int h = it++;
      if (!hok[h]) continue;
      const int good = hcnt[h];
      if (good > (best > PN_MODEL - 1 ? best : PN_MODEL - 1)) {
        best = good; best_h = h;
        const double ep = (double)(P - good) / (double)P;
        double num = fmax(1.0 - 0.99, 2.2250738585072014e-308), den = 1.0 - pow(1.0 - ep, (double)PN_MODEL);
        if (den < 2.2250738585072014e-308) niters = 0;
        else {
          num = log(num); den = log(den);
          niters = (den >= 0 || -num >= niters * (-den)) ? niters : (int)rint(num / den);
        }
      }
    }
    s_best = best_h; s_used = it;
  }
  __syncthreads();
  const int best_h = s_best;
  // ---- final EPnP over the inliers of the kept hypothesis (solvepnp.cpp), thread 0; inlier list reuses l2r ----
  if (t == 0 && best_h >= 0) {
    int n_in = 0;
    for (int p = 0; p < P; ++p) if (sq_err(hyp[best_h], p) <= 9.0) l2r[n_in++] = p;
    double R[9], tv[3];
    const bool ok = epnp(D, l2r, n_in, R, tv);
    if (!ok) s_best = -1;
    for (int i = 0; i < 9; ++i) s_R[i] = R[i];
    for (int i = 0; i < 3; ++i) s_t[i] = tv[i];
    r2l[0] = n_in;
  }
  __syncthreads();
  const bool ransac_ok = s_best >= 0;
  // ---- cv2.solvePnPRefineVVS over ALL points: Gauss-Newton on SE(3), lambda 1, <= 20 iterations, 1e-6 on the residual change ----
  double prev = 1e300;
  for (int iter = 0; ransac_ok && iter < 20; ++iter) {
    double acc[27];
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    double e2 = 0.0;
    for (int p = t; p < P; p += PN_T) {
      const double X = ((s_R[0] * wx[p] + s_R[1] * wy[p]) + s_R[2] * wz[p]) + s_t[0];
      const double Y = ((s_R[3] * wx[p] + s_R[4] * wy[p]) + s_R[5] * wz[p]) + s_t[1];
      const double Z = ((s_R[6] * wx[p] + s_R[7] * wy[p]) + s_R[8] * wz[p]) + s_t[2];
      const double x = X / Z, y = Y / Z;
      const double ex = x - (p1u[p] - D.uc) / D.fu, ey = y - (p1v[p] - D.vc) / D.fv;
      e2 += ex * ex + ey * ey;
      const double Lx[6] = {-1 / Z, 0.0, x / Z, x * y, -(1 + x * x), y}, Ly[6] = {0.0, -1 / Z, y / Z, 1 + y * y, -x * y, -x};
      int k = 0;
      for (int a = 0; a < 6; ++a) for (int c = a; c < 6; ++c) acc[k++] += Lx[a] * Lx[c] + Ly[a] * Ly[c];      // L^T L (21)
      for (int a = 0; a < 6; ++a) acc[21 + a] += Lx[a] * ex + Ly[a] * ey;                                     // L^T e (6)
    }
    double sum[27];
    for (int i = 0; i < 27; ++i) sum[i] = pn_block_sum(acc[i], red);
    const double err = sqrt(pn_block_sum(e2, red) / P);
    if (fabs(err - prev) < 1e-6) break;                    // every thread sees the same sums: uniform exit
    prev = err;
    if (t == 0) {
      double A[36], g[6], dq[6];
      int k = 0;
      for (int a = 0; a < 6; ++a) for (int c = a; c < 6; ++c) { A[a * 6 + c] = sum[k]; A[c * 6 + a] = sum[k]; ++k; }
      for (int a = 0; a < 6; ++a) g[a] = sum[21 + a];
      const bool ok = ls_solve(A, g, 6, 6, dq);              // pinv(L) e = (L^T L)^-1 L^T e for a full-rank L
      if (ok) {
        for (int a = 0; a < 6; ++a) dq[a] = -dq[a];
        const double w[3] = {dq[3], dq[4], dq[5]};
        const double th = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
        double Rw[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Vm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (th >= 1e-12) {
          const double kx = w[0] / th, ky = w[1] / th, kz = w[2] / th;
          const double Kx[9] = {0, -kz, ky, kz, 0, -kx, -ky, kx, 0};
          double K2[9];
          for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) K2[i * 3 + j] = Kx[i * 3] * Kx[j] + Kx[i * 3 + 1] * Kx[3 + j] + Kx[i * 3 + 2] * Kx[6 + j];
          const double sn = sin(th), cs = cos(th);
          for (int i = 0; i < 9; ++i) {
            Rw[i] = (i % 4 == 0 ? 1.0 : 0.0) + sn * Kx[i] + (1 - cs) * K2[i];
            Vm[i] = (i % 4 == 0 ? 1.0 : 0.0) + (1 - cs) / th * Kx[i] + (1 - sn / th) * K2[i];
          }
        }
        double dt[3], Rn[9], tn[3];
        for (int i = 0; i < 3; ++i) dt[i] = Vm[i * 3] * dq[0] + Vm[i * 3 + 1] * dq[1] + Vm[i * 3 + 2] * dq[2];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = Rw[0 * 3 + i] * s_R[0 * 3 + j] + Rw[1 * 3 + i] * s_R[1 * 3 + j] + Rw[2 * 3 + i] * s_R[2 * 3 + j];
        for (int i = 0; i < 3; ++i) tn[i] = Rw[0 * 3 + i] * (s_t[0] - dt[0]) + Rw[1 * 3 + i] * (s_t[1] - dt[1]) + Rw[2 * 3 + i] * (s_t[2] - dt[2]);
        for (int i = 0; i < 9; ++i) s_R[i] = Rn[i];
        for (int i = 0; i < 3; ++i) s_t[i] = tn[i];
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    const bool ok = have_scale && ransac_ok;
    double* o = srt + (long long)b * 13;
    o[0] = scale;
    for (int i = 0; i < 9; ++i) o[1 + i] = ok ? s_R[i] : (i % 4 == 0 ? 1.0 : 0.0);
    for (int i = 0; i < 3; ++i) o[10 + i] = ok ? s_t[i] : 0.0;
    int* io = info + (long long)b * 4;
    io[0] = nm; io[1] = ransac_ok ? 1 : 0; io[2] = ransac_ok ? r2l[0] : 0; io[3] = s_used;
    double Rf[9];
    for (int i = 0; i < 9; ++i) Rf[i] = (double)(float)s_R[i];
    const float tf[3] = {(float)s_t[0], (float)s_t[1], (float)s_t[2]};
    const double size[3] = {2.0 * (double)hmax[0] * scale, 2.0 * (double)hmax[1] * scale, 2.0 * (double)hmax[2] * scale};
    emit_bbox_world(b, Rf, tf, size, ok, E1in, bbox, valid);
  }
}

int launch_pnp_ransac(const float* nocs1, const float* pts1, const float* nocs2, const float* pts2, const double* K, const double* E1,
                      const double* E2, double* bbox, double* srt, int* info, int* valid, int B, int P, unsigned seed, hipStream_t s) {
  RGBM_REQUIRE(nocs1 && pts1 && nocs2 && pts2 && K && E1 && E2 && bbox && srt && info && valid, "pnp_ransac arguments");
  RGBM_REQUIRE(B > 0 && P >= 8 && P <= PN_P, "pnp_ransac needs 8 <= P <= 1024");
  hipLaunchKernelGGL(pnp_ransac_kernel, dim3(B), dim3(PN_T), 0, s, nocs1, pts1, nocs2, pts2, K, E1, E2, bbox, srt, info, valid, P, seed);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
