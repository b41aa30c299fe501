// Tail of AdaPoseEstimator_v5.predict shared by both post-processing kernels (postproc.hip, align.hip):
//   bbox corners of `size` (lib/utils.py:49-56), transformed by sRT = [R | t] held in float32 (interface_v5.py:357-361,
//   utils.py:58-74), then taken to the world frame with inv(view1_extrinsic), or default_bbox when anything is non-finite
//   (interface_v5.py:368-374).  Called by one thread per pose.
#pragma once
#include <hip/hip_runtime.h>

namespace rgbm {

__device__ inline void emit_bbox_world(long long b, const double R[9] /* float32 values */, const float tf[3], const double size[3],
                                       bool ok, const double* __restrict__ E1, double* __restrict__ bbox, int* __restrict__ valid) {
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { a[i][j] = E1[b * 16 + i * 4 + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {
    int piv = c; double best = fabs(a[c][c]);
    for (int rr = c + 1; rr < 4; ++rr) if (fabs(a[rr][c]) > best) { best = fabs(a[rr][c]); piv = rr; }
    if (!(best > 0.0)) { ok = false; break; }
    if (piv != c) for (int k = 0; k < 8; ++k) { const double tmp = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = tmp; }
    const double inv = 1.0 / a[c][c];
    for (int k = 0; k < 8; ++k) a[c][k] *= inv;
    for (int rr = 0; rr < 4; ++rr) if (rr != c) { const double f = a[rr][c]; for (int k = 0; k < 8; ++k) a[rr][k] -= f * a[c][k]; }
  }
  for (int i = 0; i < 4 && ok; ++i) for (int j = 0; j < 4; ++j) if (!isfinite(a[i][4 + j])) ok = false;
  double out[8][3];
  const double sg[8][3] = {{1, 1, 1}, {1, 1, -1}, {-1, 1, 1}, {-1, 1, -1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, 1}, {-1, -1, -1}};
  for (int k = 0; k < 8 && ok; ++k) {
    const double p[3] = {sg[k][0] * size[0] / 2, sg[k][1] * size[1] / 2, sg[k][2] * size[2] / 2};
    double cam[3];
    for (int i = 0; i < 3; ++i) cam[i] = R[i * 3 + 0] * p[0] + R[i * 3 + 1] * p[1] + R[i * 3 + 2] * p[2] + (double)tf[i];
    for (int i = 0; i < 3; ++i) {
      if (!isfinite(cam[i])) ok = false;
      out[k][i] = a[i][4] * cam[0] + a[i][5] * cam[1] + a[i][6] * cam[2] + a[i][7];
    }
  }
  const double dflt[8][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {0, 1, 1}, {1, 0, 0}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
  for (int k = 0; k < 8; ++k)
    for (int i = 0; i < 3; ++i) bbox[(b * 8 + k) * 3 + i] = ok ? out[k][i] : dflt[k][i] + 10.0;
  valid[b] = ok ? 1 : 0;
}

}  // namespace rgbm
