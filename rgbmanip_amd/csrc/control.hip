// Step / reward half of the RL pose controller, one thread per environment, float64 like the reference's
// numpy host code (built with -ffp-contract=off so every sum is evaluated in numpy's order without fused multiply-adds):
//   control_action_kernel      action -> camera target pose       models/controller/rl_pose.py:390-408 + utils/transform.py:50-99
//   control_reward_kernel      the 14 reward terms + 3 loss logs  models/controller/rl_pose.py:225-358
//   control_grasp_frame_kernel centre / axes handed to the manipulation planner   rl_pose.py:364-377
// The reference runs these as numpy expressions over [num_envs] arrays between two host<->device copies of every frame;
// here they read the device-resident queues directly, so a controller step never leaves the GPU.
#include "common.h"
#include "kernels.h"
#include "control.h"

namespace rgbm {

namespace {

__device__ __forceinline__ double norm3(double a, double b, double c) { return sqrt(a * a + b * b + c * c); }
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }   // np.clip (NaN stays NaN)
__device__ __forceinline__ float clipf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// Quaternion (w, x, y, z) of the rotation whose columns are the orthonormal, right-handed axes x, y, z: what the largest
// eigenvector of Horn's matrix (utils/transform.py:168-211) is for an exact frame.  The eigenvector's sign is LAPACK's
// choice in the reference (q and -q are one rotation); here the first non-negligible component is made positive.
__device__ void frame_to_quat(const double x[3], const double y[3], const double z[3], double q[4]) {
  const double r00 = x[0], r10 = x[1], r20 = x[2], r01 = y[0], r11 = y[1], r21 = y[2], r02 = z[0], r12 = z[1], r22 = z[2];
  const double tr = r00 + r11 + r22;
  if (tr > 0.0) {
    const double s = sqrt(tr + 1.0) * 2.0;
    q[0] = 0.25 * s; q[1] = (r21 - r12) / s; q[2] = (r02 - r20) / s; q[3] = (r10 - r01) / s;
  } else if (r00 > r11 && r00 > r22) {
    const double s = sqrt(1.0 + r00 - r11 - r22) * 2.0;
    q[0] = (r21 - r12) / s; q[1] = 0.25 * s; q[2] = (r01 + r10) / s; q[3] = (r02 + r20) / s;
  } else if (r11 > r22) {
    const double s = sqrt(1.0 + r11 - r00 - r22) * 2.0;
    q[0] = (r02 - r20) / s; q[1] = (r01 + r10) / s; q[2] = 0.25 * s; q[3] = (r12 + r21) / s;
  } else {
    const double s = sqrt(1.0 + r22 - r00 - r11) * 2.0;
    q[0] = (r10 - r01) / s; q[1] = (r02 + r20) / s; q[2] = (r12 + r21) / s; q[3] = 0.25 * s;
  }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double sign = 1.0;
  for (int i = 0; i < 4; ++i) {
    if (fabs(q[i]) > 1e-12) { sign = q[i] < 0.0 ? -1.0 : 1.0; break; }
  }
  for (int i = 0; i < 4; ++i) q[i] = sign * q[i] / n;
}

// lookat_quat for one direction (transform.py:56-97).  `batch_zero` is the reference's whole-batch norm test (:69).
__device__ void lookat_quat_dev(const double dir[3], bool batch_zero, double q[4]) {
  const double n = norm3(dir[0], dir[1], dir[2]) + 1e-9;
  const double d[3] = {dir[0] / n, dir[1] / n, dir[2] / n};
  double x[3], y[3], z[3];
  const double dot = d[2];
  if (batch_zero) {
    x[0] = 1; x[1] = 0; x[2] = 0; y[0] = 0; y[1] = 1; y[2] = 0; z[0] = 0; z[1] = 0; z[2] = 1;
  } else if (fabs(dot + 1.0) < 1e-6) {
    x[0] = 0; x[1] = 0; x[2] = -1; y[0] = 0; y[1] = 1; y[2] = 0; z[0] = 1; z[1] = 0; z[2] = 0;
  } else if (fabs(dot - 1.0) < 1e-6) {
    x[0] = 0; x[1] = 0; x[2] = 1; y[0] = 0; y[1] = 1; y[2] = 0; z[0] = -1; z[1] = 0; z[2] = 0;
  } else {
    // y = z_ x d, z = d x y, both normalised; x = d up to its 1e-9 length defect, which Horn's best-fit rotation ignores
    const double dn = norm3(d[0], d[1], d[2]);
    x[0] = d[0] / dn; x[1] = d[1] / dn; x[2] = d[2] / dn;
    const double yn = sqrt(d[1] * d[1] + d[0] * d[0]);
    y[0] = -d[1] / yn; y[1] = d[0] / yn; y[2] = 0.0;
    double zz[3] = {x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]};
    const double zn = norm3(zz[0], zz[1], zz[2]);
    z[0] = zz[0] / zn; z[1] = zz[1] / zn; z[2] = zz[2] / zn;
  }
  frame_to_quat(x, y, z, q);
}

}  // namespace

// directions [N,3] f64 -> quaternions [N,4] f64 (utils.transform.lookat_quat)
__global__ void lookat_quat_kernel(const double* dir, double* quat, int N, int batch_zero) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double d[3] = {dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2]};
  double q[4];
  lookat_quat_dev(d, batch_zero != 0, q);
  for (int k = 0; k < 4; ++k) quat[i * 4 + k] = q[k];
}

// action [N,lda] f32 -> pose [N,7] f64: xyz = clip(a[:3] + mid, min, max); heading (1, dy, dz) -> quaternion
__global__ void control_action_kernel(const float* action, int lda, double3 mid, double3 lo, double3 hi, double* pose, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float* a = action + (long long)i * lda;
  const double ln = 1.0 / (1.0 + 1e-9);                              // heading / (|heading| + 1e-9), heading = (1,0,0)
  const double dir[3] = {ln, ln * (double)a[3], (double)a[4]};      // lookat_norm + (z_ x lookat_norm) * dy + z_ * dz
  double q[4];
  lookat_quat_dev(dir, false, q);
  double* p = pose + (long long)i * 7;
  p[0] = clipd((double)a[0] + mid.x, lo.x, hi.x);
  p[1] = clipd((double)a[1] + mid.y, lo.y, hi.y);
  p[2] = clipd((double)a[2] + mid.z, lo.z, hi.z);
  p[3] = q[0]; p[4] = q[1]; p[5] = q[2]; p[6] = q[3];
}

__global__ void control_reward_kernel(const ControlRewardArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = a.N;
  if (i >= N) return;
  const float* act = a.action + (long long)i * a.lda;
  // ---- float32 terms (the action arrives as a float32 numpy array: rl_pose.py:388) ----
  float s32 = 0.f;
  for (int k = 0; k < a.T; ++k) s32 += act[6 + k] * act[6 + k];
  const float vn = sqrtf(s32) - 1.f;
  const float view_norm_pen = clipf(vn * vn, -1.f, 1.f) * (float)a.coef[12];
  const float l0 = act[3] - act[0], l1 = act[4] - act[1], l2 = act[5] - act[2];
  const float ll = sqrtf(l0 * l0 + l1 * l1 + l2 * l2) - 1.f;
  const float xyz_lookat = clipf(ll * ll, -2.f, 2.f) * (float)a.coef[5];
  const float move_success = a.move_success[i] * (float)a.coef[1];
  // ---- camera pose terms ----
  const double* cp = a.cam_pose + (long long)i * 7;
  const double* tg = a.target + (long long)i * 7;
  double s = 0.0;
  for (int k = 0; k < 7; ++k) { const double d = cp[k] - tg[k]; s += d * d; }
  const double diff = clipd(sqrt(s), -2.0, 2.0) * a.coef[0];
  const double far = clipd(norm3(cp[0] - a.proper_pos[0], cp[1] - a.proper_pos[1], cp[2] - a.proper_pos[2]), -2.0, 2.0) * a.coef[3];
  // quat_to_axis(cam_pose[:, 3:], 0) with the reference's batch scramble (transform.py:234): element 3i+j of [A.., B.., C..]
  double ori[3];
  for (int j = 0; j < 3; ++j) {
    const int m = 3 * i + j, comp = m / N, e = m - comp * N;
    const double* q = a.cam_pose + (long long)e * 7 + 3;
    ori[j] = comp == 0 ? 2 * (q[0] * q[0]) + 2 * (q[1] * q[1]) - 1 : comp == 1 ? 2 * q[1] * q[2] + 2 * q[0] * q[3]
                                                                               : 2 * q[1] * q[3] - 2 * q[0] * q[2];
  }
  // ---- image-space bbox terms ----
  const double* bb = a.bbox + (long long)i * 4;
  const double avail = a.avail[i] != 0.0 ? 1.0 : 0.0;
  const double bx = (bb[0] + bb[2]) / 2 - 0.5, by = (bb[1] + bb[3]) / 2 - 0.5;
  const double bbox_pen = clipd(sqrt(bx * bx + by * by) * avail, -1.0, 1.0) * a.coef[6];
  const bool edge = (bb[0] <= 1e-9) || (bb[1] <= 1e-9) || (bb[2] >= 1 - 1e-9) || (bb[3] >= 1 - 1e-9);
  const float bbox_edge = (edge ? 1.f : 0.f) * (float)a.coef[7];
  const double have_bbox = avail * a.coef[8];
  // ---- pose-estimate terms ----
  const double* g = a.gt_bbox + (long long)i * 24;
  const double* p = a.pred_bbox + (long long)i * 24;
  double gc[3], go[3], pc[3], po[3];
  for (int k = 0; k < 3; ++k) {
    gc[k] = (g[k] + g[18 + k]) / 2;  go[k] = g[k] - g[12 + k];
    pc[k] = (p[k] + p[21 + k]) / 2;  po[k] = p[3 + k] - p[k];
  }
  const double gn = norm3(go[0], go[1], go[2]) + 1e-9, pn = norm3(po[0], po[1], po[2]) + 1e-9;
  double cd[3], od[3];
  for (int k = 0; k < 3; ++k) { cd[k] = pc[k] - gc[k]; od[k] = po[k] / pn - go[k] / gn; }
  if (a.pots) { cd[0] *= 3; cd[1] *= 3; }
  const double center_diff = clipd(norm3(cd[0], cd[1], cd[2]), -20.0, 20.0);
  const double open_diff = clipd(norm3(od[0], od[1], od[2]) * 2, -20.0, 20.0);
  double center_rew = a.precision2 / (a.precision2 + center_diff * center_diff);
  double open_rew = 1 / (1 + open_diff * open_diff);
  // ---- viewpoint terms ----
  const double* rr = a.robot_pose + (long long)i * 7;
  const double* pq = a.pose_cur + (long long)i * 7;
  const double* pl = a.pose_prev + (long long)i * 7;
  double to[3], lv[3], tv[3];
  for (int k = 0; k < 3; ++k) {
    to[k] = gc[k] - (rr[k] + pq[k]);
    const double rel = gc[k] - rr[k];
    lv[k] = pl[k] - rel; tv[k] = pq[k] - rel;
  }
  const double tn = norm3(to[0], to[1], to[2]) + 1e-9, ln = norm3(lv[0], lv[1], lv[2]) + 1e-9, wn = norm3(tv[0], tv[1], tv[2]) + 1e-9;
  const double ori_rew = (ori[0] * (to[0] / tn) + ori[1] * (to[1] / tn) + ori[2] * (to[2] / tn)) * a.coef[4];
  const double move_period = norm3(pl[0] - pq[0], pl[1] - pq[1], pl[2] - pq[2]) * a.coef[2];
  double view_rew = 0.0;
  if (a.first) { center_rew *= 0; open_rew *= 0; }
  else {
    const double c = (lv[0] / ln) * (tv[0] / wn) + (lv[1] / ln) * (tv[1] / wn) + (lv[2] / ln) * (tv[2] / wn);
    view_rew = acos(c) > 0.3 ? 1.0 : 0.0;                            // NaN (|c| > 1 by rounding) compares false, like np.where
  }
  center_rew *= a.coef[9]; open_rew *= a.coef[10]; view_rew *= a.coef[11];
  const double success = a.success[i] * a.coef[13];
  const double reward = diff + (double)move_success + move_period + far + ori_rew + (double)xyz_lookat + bbox_pen + (double)bbox_edge
                        + have_bbox + center_rew + open_rew + view_rew + (double)view_norm_pen + success;
  a.reward[i] = reward;
  if (a.terms) {
    const double t[17] = {diff, (double)move_success, move_period, far, ori_rew, (double)xyz_lookat, bbox_pen, (double)bbox_edge,
                          have_bbox, center_rew, open_rew, view_rew, (double)view_norm_pen, success, center_diff, open_diff,
                          far};                                      // LOSS:far aliases the scaled far term (rl_pose.py:247, 322)
    for (int k = 0; k < 17; ++k) a.terms[(long long)k * N + i] = t[k];
  }
}

// est [N,8,3] f64 -> center [N,3], direction [N,3,3] (rows: b1-b0, b0-b2, b4-b0 normalised; identity rows where degenerate)
__global__ void control_grasp_frame_kernel(const double* est, double* center, double* direction, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double* b = est + (long long)i * 24;
  for (int k = 0; k < 3; ++k) center[i * 3 + k] = (b[k] + b[21 + k]) / 2;
  double d[3][3];
  for (int k = 0; k < 3; ++k) { d[0][k] = b[3 + k] - b[k]; d[1][k] = b[k] - b[6 + k]; d[2][k] = b[12 + k] - b[k]; }
  for (int r = 0; r < 3; ++r) {
    const double n = norm3(d[r][0], d[r][1], d[r][2]);
    for (int k = 0; k < 3; ++k) direction[i * 9 + r * 3 + k] = n > 1e-8 ? d[r][k] / (n + 1e-8) : (r == k ? 1.0 : 0.0);
  }
}

int launch_lookat_quat(const double* dir, double* quat, int N, int batch_zero, hipStream_t s) {
  RGBM_REQUIRE(dir && quat && N > 0, "lookat_quat arguments");
  hipLaunchKernelGGL(lookat_quat_kernel, dim3((N + 127) / 128), dim3(128), 0, s, dir, quat, N, batch_zero);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_control_action(const float* action, int lda, const double* pose_mid, const double* pose_min, const double* pose_max,
                          double* pose, int N, hipStream_t s) {
  RGBM_REQUIRE(action && pose_mid && pose_min && pose_max && pose && N > 0 && lda >= 5, "control_action arguments");
  hipLaunchKernelGGL(control_action_kernel, dim3((N + 127) / 128), dim3(128), 0, s, action, lda,
                     make_double3(pose_mid[0], pose_mid[1], pose_mid[2]), make_double3(pose_min[0], pose_min[1], pose_min[2]),
                     make_double3(pose_max[0], pose_max[1], pose_max[2]), pose, N);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_control_reward(const ControlRewardArgs& a, hipStream_t s) {
  RGBM_REQUIRE(a.action && a.cam_pose && a.target && a.move_success && a.bbox && a.avail && a.gt_bbox && a.pred_bbox &&
               a.pose_cur && a.pose_prev && a.robot_pose && a.success && a.reward, "control_reward arguments");
  RGBM_REQUIRE(a.N > 0 && a.T > 0 && a.lda >= 6 + a.T, "control_reward sizes");
  hipLaunchKernelGGL(control_reward_kernel, dim3((a.N + 127) / 128), dim3(128), 0, s, a);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_control_grasp_frame(const double* est, double* center, double* direction, int N, hipStream_t s) {
  RGBM_REQUIRE(est && center && direction && N > 0, "control_grasp_frame arguments");
  hipLaunchKernelGGL(control_grasp_frame_kernel, dim3((N + 127) / 128), dim3(128), 0, s, est, center, direction, N);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
