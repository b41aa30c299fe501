// Device-side AdaPoseEstimator_v5.prepare_model_input, batched over frames:
//   /root/reference/models/pose_estimator/AdaPose/interface_v5.py:58-170  (mask bbox -> crop window -> resized mask ->
//     1024 `choose` indices -> resized + normalised RGB -> crop-adjusted intrinsics)
//   /root/reference/models/pose_estimator/AdaPose/lib/utils.py:10-38      (get_bbox: square window, multiple of 40, <= 440)
// The reference does this per sample on the host with numpy + cv2.resize (INTER_NEAREST for the mask, INTER_LINEAR for the
// float image); OpenCV is not available in the build container, so the resize arithmetic is OpenCV's documented one, the
// same as in oracle/postproc_ref.py (half-pixel centres, edge clamp, no antialias; sx = floor(dx*scale) for nearest) —
// "parity unpinned" at exactly this step, pinned (bit-exact) against the oracle restatement.
// The reference draws the 1024-subset with the global np.random.shuffle; here it is a seeded hash: candidate i of frame f
// gets key = mix32(seed, f, i) and the 1024 smallest (key, i) pairs are kept in index order — a uniformly random ordered
// subset like the reference's, but reproducible (oracle: postproc_ref.choose_subset_hash).
// Built with -ffp-contract=off: numpy rounds every elementwise op separately.
#include "common.h"
#include "kernels.h"

namespace rgbm {

namespace {

constexpr int PRE_THREADS = 1024;

__host__ __device__ inline unsigned mix32(unsigned seed, unsigned frame, unsigned idx) {
  unsigned h = seed ^ (frame * 0x9E3779B9u) ^ (idx * 0x85EBCA6Bu);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;     // murmur3 finaliser
  return h;
}

// ---- 1. mask bounding box -> crop window (get_bbox) -> cropped intrinsics ------------------------------------------
// frame_map (optional): frame f reads image / mask number frame_map[f] of the arrays it is given (a view queue); a negative
// entry is a missing view and behaves like an all-zero mask.  K, the outputs and the subset hash stay indexed by f.
__global__ __launch_bounds__(PRE_THREADS) void mask_window_kernel(const unsigned char* __restrict__ mask, const double* __restrict__ K,
                                                                   const int* __restrict__ frame_map,
                                                                   int H, int W, int S, int* __restrict__ window /*[N,4]*/,
                                                                   double* __restrict__ Kcrop /*[N,9]*/, int* __restrict__ valid) {
  __shared__ int s_y1, s_x1, s_y2, s_x2;
  const int f = blockIdx.x, t = threadIdx.x;
  if (t == 0) { s_y1 = H; s_x1 = W; s_y2 = -1; s_x2 = -1; }
  __syncthreads();
  const int sf = frame_map ? frame_map[f] : f;
  const unsigned char* m = mask + (long long)(sf < 0 ? 0 : sf) * H * W;
  int y1 = H, x1 = W, y2 = -1, x2 = -1;
  for (int i = t; sf >= 0 && i < H * W; i += PRE_THREADS) {
    if (m[i]) {
      const int y = i / W, x = i - y * W;
      y1 = min(y1, y); y2 = max(y2, y); x1 = min(x1, x); x2 = max(x2, x);
    }
  }
  if (y2 >= 0) { atomicMin(&s_y1, y1); atomicMax(&s_y2, y2); atomicMin(&s_x1, x1); atomicMax(&s_x2, x2); }
  __syncthreads();
  if (t != 0) return;
  int* w = window + f * 4;
  double* Ko = Kcrop + f * 9;
  if (s_y2 < 0) {                                   // empty mask: the reference returns None -> default bbox
    w[0] = 0; w[1] = 40; w[2] = 0; w[3] = 40;
    for (int i = 0; i < 9; ++i) Ko[i] = (i % 4 == 0) ? 1.0 : 0.0;
    valid[f] = 0;
    return;
  }
  // lib/utils.py:10-38 (integer arithmetic of the reference; image fixed at 480 x 640 there, H x W here)
  int win = (max(s_y2 - s_y1, s_x2 - s_x1) / 40 + 1) * 40;
  win = min(win, 440);
  const int cy = (s_y1 + s_y2) / 2, cx = (s_x1 + s_x2) / 2;
  int rmin = cy - win / 2, rmax = cy + win / 2, cmin = cx - win / 2, cmax = cx + win / 2;
  if (rmin < 0) { rmax = rmax - rmin; rmin = 0; }
  if (cmin < 0) { cmax = cmax - cmin; cmin = 0; }
  if (rmax > H) { rmin = rmin - (rmax - H); rmax = H; }
  if (cmax > W) { cmin = cmin - (cmax - W); cmax = W; }
  w[0] = rmin; w[1] = rmax; w[2] = cmin; w[3] = cmax;
  // interface_v5.py:153-168 (python floats = fp64)
  const double ratio = (double)S / (double)(rmax - rmin);
  const double* Ki = K + f * 9;
  const double ccx = (double)(cmin + cmax) / 2.0, ccy = (double)(rmin + rmax) / 2.0;
  const double csx = (double)(cmax - cmin + 1), csy = (double)(rmax - rmin + 1);
  Ko[0] = Ki[0] * ratio; Ko[1] = 0.0; Ko[2] = (Ki[2] - (ccx - csx / 2.0)) * ratio;
  Ko[3] = 0.0; Ko[4] = Ki[4] * ratio; Ko[5] = (Ki[5] - (ccy - csy / 2.0)) * ratio;
  Ko[6] = 0.0; Ko[7] = 0.0; Ko[8] = 1.0;
  valid[f] = 1;
}

// ---- 2. crop + resize: nearest mask, bilinear RGB, ToTensor + Normalize ---------------------------------------------
__global__ void crop_resize_kernel(const float* __restrict__ rgb /*[N,H,W,3]*/, const unsigned char* __restrict__ mask,
                                   const int* __restrict__ frame_map, const int* __restrict__ window, int N, int H, int W, int S,
                                   float* __restrict__ img /*[N,3,S,S]*/, unsigned char* __restrict__ small /*[N,S,S]*/) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * S * S) return;
  const int f = (int)(i / (S * S)), p = (int)(i - (long long)f * S * S), dy = p / S, dx = p - dy * S;
  const int rmin = window[f * 4 + 0], rmax = window[f * 4 + 1], cmin = window[f * 4 + 2], cmax = window[f * 4 + 3];
  const int h = rmax - rmin, w = cmax - cmin;
  // INTER_NEAREST: sx = min(floor(dx * (src/dst)), src-1), fp64 like numpy's python-float scale
  const int ny = min((int)floor((double)dy * ((double)h / (double)S)), h - 1);
  const int nx = min((int)floor((double)dx * ((double)w / (double)S)), w - 1);
  const int mf = frame_map ? frame_map[f] : f;
  const int sf = mf < 0 ? 0 : mf;
  small[i] = (mf >= 0 && mask[((long long)sf * H + rmin + ny) * W + cmin + nx]) ? 1 : 0;
  // INTER_LINEAR on a float image: f = (d + 0.5) * scale - 0.5, clamp at both edges, weights in fp32
  auto taps = [](int d, int n_src, int n_dst, int& i0, int& i1, float& a) {
    const double fl = ((double)d + 0.5) * ((double)n_src / (double)n_dst) - 0.5;
    int j = (int)floor(fl);
    a = (float)(fl - (double)j);
    if (j < 0) { j = 0; a = 0.f; }
    if (j >= n_src - 1) { i0 = n_src - 1; i1 = n_src - 1; a = 0.f; }
    else { i0 = j; i1 = j + 1; }
  };
  int y0, y1, x0, x1;
  float ay, ax;
  taps(dy, h, S, y0, y1, ay);
  taps(dx, w, S, x0, x1, ax);
  const float* base = rgb + (long long)sf * H * W * 3;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float p00 = base[((long long)(rmin + y0) * W + cmin + x0) * 3 + c], p01 = base[((long long)(rmin + y0) * W + cmin + x1) * 3 + c];
    const float p10 = base[((long long)(rmin + y1) * W + cmin + x0) * 3 + c], p11 = base[((long long)(rmin + y1) * W + cmin + x1) * 3 + c];
    const float top = p00 * (1.f - ax) + p01 * ax;
    const float bot = p10 * (1.f - ax) + p11 * ax;
    const float v = top * (1.f - ay) + bot * ay;
    img[((long long)f * 3 + c) * S * S + p] = (v - mean[c]) / stdv[c];
  }
}

// ---- 3. choose: nonzero indices of the resized mask, ordered random P-subset or wrap padding -------------------------
__global__ __launch_bounds__(PRE_THREADS) void choose_kernel(const unsigned char* __restrict__ small, const int* __restrict__ window,
                                                              int S, int P, unsigned seed, int* __restrict__ choose /*[N,P]*/,
                                                              float* __restrict__ pts2d /*[N,P,2] or null*/, int* __restrict__ valid,
                                                              int frame0 /*hash index of frame 0: a batch prepared in pieces*/) {
  extern __shared__ unsigned short idx[];           // candidate pixel indices, in order (S*S <= 65536)
  __shared__ int wsum[PRE_THREADS / 64];
  __shared__ int s_total, s_cnt;
  const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const unsigned char* m = small + (long long)f * S * S;
  const int per = (S * S + PRE_THREADS - 1) / PRE_THREADS;
  const int lo = t * per, hi = min(lo + per, S * S);

  // block-wide exclusive scan of a per-thread count, thread ranges are contiguous so order is preserved
  auto excl_scan = [&](int mine, int& total) {
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    __syncthreads();                                // wsum free to overwrite
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int k = 0; k < PRE_THREADS / 64; ++k) { if (k < wv) base += wsum[k]; tot += wsum[k]; }
    total = tot;
    return base + incl - mine;
  };

  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += m[i] ? 1 : 0;
  int n;
  int pos = excl_scan(cnt, n);
  for (int i = lo; i < hi; ++i) if (m[i]) idx[pos++] = (unsigned short)i;
  __syncthreads();
  int* out = choose + (long long)f * P;
  if (n == 0 || !valid[f]) {                        // reference: None -> default bbox; keep the tensors finite
    for (int i = t; i < P; i += PRE_THREADS) out[i] = 0;
    if (t == 0) valid[f] = 0;
  } else if (n <= P) {
    for (int i = t; i < P; i += PRE_THREADS) out[i] = idx[i % n];                   // np.pad(choose, ..., "wrap")
  } else {
    // smallest threshold T with count(key <= T) >= P, by bisection over the 32-bit key space
    unsigned tlo = 0u, thi = 0xFFFFFFFFu;
    while (tlo < thi) {
      const unsigned mid = tlo + ((thi - tlo) >> 1);
      int c = 0;
      for (int i = t; i < n; i += PRE_THREADS) c += mix32(seed, (unsigned)(f + frame0), (unsigned)idx[i]) <= mid ? 1 : 0;
      if (t == 0) s_cnt = 0;
      __syncthreads();
      atomicAdd(&s_cnt, c);
      __syncthreads();
      const int tot = s_cnt;
      __syncthreads();
      if (tot >= P) thi = mid; else tlo = mid + 1;
    }
    const unsigned T = tlo;
    // keep every key < T and, in index order, as many key == T as still fit; ordered compaction (contiguous ranges per thread)
    const int per2 = (n + PRE_THREADS - 1) / PRE_THREADS;
    const int a = min(t * per2, n), b = min(a + per2, n);
    int c_lt = 0, c_eq = 0;
    for (int i = a; i < b; ++i) { const unsigned k = mix32(seed, (unsigned)(f + frame0), (unsigned)idx[i]); c_lt += k < T; c_eq += k == T; }
    int n_lt, n_eq;
    int p_lt = excl_scan(c_lt, n_lt);
    int p_eq = excl_scan(c_eq, n_eq);
    const int eq_keep = P - n_lt;                   // ties kept (>= 1)
    // output position of element i = (#kept with smaller index): kept_lt before + min(eq before, eq_keep)
    for (int i = a; i < b; ++i) {
      const unsigned k = mix32(seed, (unsigned)(f + frame0), (unsigned)idx[i]);
      if (k < T) { out[p_lt + min(p_eq, eq_keep)] = idx[i]; ++p_lt; }
      else if (k == T) { if (p_eq < eq_keep) out[p_lt + p_eq] = idx[i]; ++p_eq; }
    }
    (void)n_eq;
  }
  if (pts2d) {                                      // interface_v5.py:140-147 (float32 / python-float ratio, then + int)
    __syncthreads();
    const int rmin = window[f * 4 + 0], rmax = window[f * 4 + 1], cmin = window[f * 4 + 2];
    const float ratio = (float)((double)S / (double)(rmax - rmin));     // float32 array / python float stays float32 in numpy
    for (int i = t; i < P; i += PRE_THREADS) {
      const int c = out[i];
      pts2d[((long long)f * P + i) * 2 + 0] = (float)(c % S) / ratio + (float)cmin;
      pts2d[((long long)f * P + i) * 2 + 1] = (float)(c / S) / ratio + (float)rmin;
    }
  }
}

// ---- ControlInterface.add_view (rl_pose.py:130-149): per-env mask extent, in rows (dim 1) and columns (dim 2) -----------
__global__ __launch_bounds__(PRE_THREADS) void mask_extent_kernel(const unsigned char* __restrict__ mask, int H, int W,
                                                                   int* __restrict__ ext /*[N,4] rmin,cmin,rmax,cmax*/,
                                                                   int* __restrict__ count /*[N]*/) {
  __shared__ int s_y1, s_x1, s_y2, s_x2, s_n;
  const int f = blockIdx.x, t = threadIdx.x;
  if (t == 0) { s_y1 = 2 * H; s_x1 = 2 * W; s_y2 = 0; s_x2 = 0; s_n = 0; }      // the reference's defaults for "no pixel"
  __syncthreads();
  const unsigned char* m = mask + (long long)f * H * W;
  int y1 = 2 * H, x1 = 2 * W, y2 = 0, x2 = 0, n = 0;
  for (int i = t; i < H * W; i += PRE_THREADS) {
    if (m[i]) {
      const int y = i / W, x = i - y * W;
      y1 = min(y1, y); y2 = max(y2, y); x1 = min(x1, x); x2 = max(x2, x); ++n;
    }
  }
  if (n) { atomicMin(&s_y1, y1); atomicMax(&s_y2, y2); atomicMin(&s_x1, x1); atomicMax(&s_x2, x2); atomicAdd(&s_n, n); }
  __syncthreads();
  if (t == 0) { ext[f * 4 + 0] = s_y1; ext[f * 4 + 1] = s_x1; ext[f * 4 + 2] = s_y2; ext[f * 4 + 3] = s_x2; count[f] = s_n; }
}

}  // namespace

// P = eye(4); P[:3, :] = K' @ E[:3, :] in fp64 (interface_v5.py:264-267: numpy float64, dot products summed left to right), stored
// as the float32 the network takes (interface_v5.py:269-270).  One thread per matrix.
__global__ void projection_kernel(const double* __restrict__ Kc, const double* __restrict__ E, float* __restrict__ P, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* k = Kc + (size_t)i * 9;
  const double* e = E + (size_t)i * 16;
  float* p = P + (size_t)i * 16;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) p[r * 4 + c] = (float)((k[r * 3 + 0] * e[c] + k[r * 3 + 1] * e[4 + c]) + k[r * 3 + 2] * e[8 + c]);
  p[12] = 0.f; p[13] = 0.f; p[14] = 0.f; p[15] = 1.f;
}

int launch_projection(const double* Kc, const double* E, float* P, int n, hipStream_t s) {
  RGBM_REQUIRE(Kc && E && P && n > 0, "projection arguments");
  hipLaunchKernelGGL(projection_kernel, dim3((n + 127) / 128), dim3(128), 0, s, Kc, E, P, n);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_mask_extent(const unsigned char* mask, int N, int H, int W, int* ext, int* count, hipStream_t s) {
  RGBM_REQUIRE(mask && ext && count && N > 0 && H > 0 && W > 0, "mask_extent arguments");
  hipLaunchKernelGGL(mask_extent_kernel, dim3(N), dim3(PRE_THREADS), 0, s, mask, H, W, ext, count);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

unsigned prepare_mix32(unsigned seed, unsigned frame, unsigned idx) { return mix32(seed, frame, idx); }

int launch_prepare_inputs(const float* rgb, const unsigned char* mask, const double* K, const int* frame_map, int N, int H, int W, int S, int P,
                          unsigned seed, float* img, int* choose, float* pts2d, double* Kcrop, int* window, int* valid,
                          unsigned char* small_scratch, hipStream_t s, int frame0) {
  RGBM_REQUIRE(rgb && mask && K && img && choose && Kcrop && window && valid && small_scratch, "prepare_inputs arguments");
  // the crop window is a square of up to 440 pixels shifted back into the frame (lib/utils.py:10-38 does it for 480 x 640): a
  // smaller frame could not hold it and the shifted window would start at a negative row / column
  RGBM_REQUIRE(N > 0 && H >= 440 && W >= 440 && S > 0 && P > 0 && S * S <= 65536, "prepare_inputs sizes (frames must be at least 440 x 440)");
  hipLaunchKernelGGL(mask_window_kernel, dim3(N), dim3(PRE_THREADS), 0, s, mask, K, frame_map, H, W, S, window, Kcrop, valid);
  const long long tot = (long long)N * S * S;
  hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, rgb, mask, frame_map, window, N, H, W, S, img, small_scratch);
  const size_t lds = (size_t)S * S * sizeof(unsigned short);
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(choose_kernel), 150 * 1024)) return rc;
  hipLaunchKernelGGL(choose_kernel, dim3(N), dim3(PRE_THREADS), lds, s, small_scratch, window, S, P, seed, choose, pts2d, valid, frame0);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
