// Procedural camera for the synthetic MultiVecEnv stand-in: there is no simulator on the GPU box, so the
// controller benchmark / tests render what `MultiVecEnv.get_image()` (env/my_vec_env.py:266, base_manipulation.py:653-687)
// would hand over — a 480x640 colour frame, the handle mask, the intrinsic and the world->camera extrinsic per environment —
// directly into device memory.  The scene is one oriented box (the "handle") in front of a patterned background; the mask is
// exactly the set of pixels whose viewing ray hits the box, so mask, ground-truth corners, K and E are geometrically consistent.
//
// Everything is float64 with a fixed evaluation order and no fused multiply-adds (-ffp-contract=off): oracle/synth_env_ref.py
// restates the same expressions in numpy and the two agree BIT FOR BIT (tests/test_gpu_control.py), which is what lets the
// tests drive the reference-shaped host code and the device code with identical frames.
#include "common.h"
#include "kernels.h"
#include "control.h"

namespace rgbm {

// Per environment: K [3,3], E [4,4] (OpenCV camera: x right, y down, z forward; X_cam = E X_world) and the ray set-up
// rays [N,12] = camera centre in box coordinates (3) + A (9, row-major) with ray direction in box coordinates
// d = A (xn, yn, 1), xn = (u - cx) / fx, yn = (v - cy) / fy.
__global__ void synth_camera_kernel(const SynthScene sc, double* K, double* E, double* rays) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sc.N) return;
  const double* cp = sc.cam_pose + (long long)i * 7;
  const double* rp = sc.robot_pose + (long long)i * 7;
  const double* bx = sc.box + (long long)i * 15;
  double qn = sqrt(cp[3] * cp[3] + cp[4] * cp[4] + cp[5] * cp[5] + cp[6] * cp[6]);
  const double w = cp[3] / qn, x = cp[4] / qn, y = cp[5] / qn, z = cp[6] / qn;
  // columns of R(q): forward f = R e_x, left l = R e_y, up u = R e_z
  const double f[3] = {1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)};
  const double l[3] = {2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)};
  const double u[3] = {2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)};
  const double p[3] = {rp[0] + cp[0], rp[1] + cp[1], rp[2] + cp[2]};            // camera centre in the world frame
  double R[3][3];                                                               // rows: right, down, forward
  for (int k = 0; k < 3; ++k) { R[0][k] = -l[k]; R[1][k] = -u[k]; R[2][k] = f[k]; }
  double* Ei = E + (long long)i * 16;
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) Ei[r * 4 + k] = R[r][k];
    Ei[r * 4 + 3] = -(R[r][0] * p[0] + R[r][1] * p[1] + R[r][2] * p[2]);
  }
  Ei[12] = 0; Ei[13] = 0; Ei[14] = 0; Ei[15] = 1;
  double* Ki = K + (long long)i * 9;
  Ki[0] = sc.fx; Ki[1] = 0; Ki[2] = sc.cx; Ki[3] = 0; Ki[4] = sc.fy; Ki[5] = sc.cy; Ki[6] = 0; Ki[7] = 0; Ki[8] = 1;
  double* ry = rays + (long long)i * 12;
  for (int a = 0; a < 3; ++a) {                                                 // box axis a (row a of the box rotation)
    const double* ax = bx + 3 + a * 3;
    ry[a] = ax[0] * (p[0] - bx[0]) + ax[1] * (p[1] - bx[1]) + ax[2] * (p[2] - bx[2]);
    for (int c = 0; c < 3; ++c)                                                 // A = Rbox * R^T : A[a][c] = axis_a . (row c of R)
      ry[3 + a * 3 + c] = ax[0] * R[c][0] + ax[1] * R[c][1] + ax[2] * R[c][2];
  }
}

// One thread per pixel; colour [N,H,W,3] f32 in [0,1], mask [N,H,W] u8.
__global__ __launch_bounds__(256) void synth_render_kernel(const SynthScene sc, const double* rays, float* color, unsigned char* mask) {
  const int env = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= sc.H * sc.W) return;
  const int v = pix / sc.W, uu = pix - v * sc.W;
  __shared__ double rs[12], hs[3];
  if (threadIdx.x < 12) rs[threadIdx.x] = rays[(long long)env * 12 + threadIdx.x];
  if (threadIdx.x >= 16 && threadIdx.x < 19) hs[threadIdx.x - 16] = sc.box[(long long)env * 15 + 12 + threadIdx.x - 16];
  __syncthreads();
  const double xn = ((double)uu - sc.cx) / sc.fx, yn = ((double)v - sc.cy) / sc.fy;
  double tmin = -INFINITY, tmax = INFINITY;
  int face = 0;
  double o[3], d[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    o[a] = rs[a];
    d[a] = (rs[3 + a * 3] * xn + rs[4 + a * 3] * yn) + rs[5 + a * 3];
    double tn, tf;
    if (d[a] != 0.0) {
      const double t1 = (-hs[a] - o[a]) / d[a], t2 = (hs[a] - o[a]) / d[a];
      tn = fmin(t1, t2); tf = fmax(t1, t2);
    } else {
      const bool inside = fabs(o[a]) <= hs[a];
      tn = inside ? -INFINITY : INFINITY; tf = INFINITY;
    }
    if (tn > tmin) { tmin = tn; face = a; }
    tmax = fmin(tmax, tf);
  }
  const bool hit = tmax >= tmin && tmin > 0.0;
  float rgb[3];
  if (hit) {
    const int a1 = face == 0 ? 1 : 0, a2 = face == 2 ? 1 : 2;
    const double p1 = o[a1] + tmin * d[a1], p2 = o[a2] + tmin * d[a2];
    const long long cell = (long long)floor(p1 * 50.0) + (long long)floor(p2 * 50.0);
    const double shade = (cell & 1) ? 1.0 : 0.6;
    const double base[3][3] = {{0.85, 0.30, 0.25}, {0.25, 0.80, 0.35}, {0.30, 0.40, 0.90}};
    for (int c = 0; c < 3; ++c) rgb[c] = (float)(base[face][c] * shade);
  } else {
    const int e = sc.env0 + env;
    const int m0 = (uu * 7 + v * 3 + e * 31) % 97, m1 = (uu * 2 + v * 5 + e * 17) % 89, m2 = ((uu >> 3) + (v >> 3) + e) % 13;
    rgb[0] = (float)(0.20 + 0.5 * ((double)m0 / 97.0));
    rgb[1] = (float)(0.25 + 0.4 * ((double)m1 / 89.0));
    rgb[2] = (float)(0.15 + 0.6 * ((double)m2 / 13.0));
  }
  const long long off = (long long)env * sc.H * sc.W + pix;
  color[off * 3] = rgb[0]; color[off * 3 + 1] = rgb[1]; color[off * 3 + 2] = rgb[2];
  mask[off] = hit ? 1 : 0;
}

int launch_synth_camera(const SynthScene& sc, double* K, double* E, double* rays, hipStream_t s) {
  RGBM_REQUIRE(sc.cam_pose && sc.robot_pose && sc.box && K && E && rays && sc.N > 0, "synth_camera arguments");
  hipLaunchKernelGGL(synth_camera_kernel, dim3((sc.N + 63) / 64), dim3(64), 0, s, sc, K, E, rays);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_synth_render(const SynthScene& sc, const double* rays, float* color, unsigned char* mask, hipStream_t s) {
  RGBM_REQUIRE(sc.box && rays && color && mask && sc.N > 0 && sc.H > 0 && sc.W > 0 && sc.N < 65536, "synth_render arguments");
  hipLaunchKernelGGL(synth_render_kernel, dim3((sc.H * sc.W + 255) / 256, sc.N), dim3(256), 0, s, sc, rays, color, mask);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
