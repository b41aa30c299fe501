// Bandwidth-bound helper kernels of the AdaPose forward (gfx950): layout conversion, max-pool,
// bilinear x2 up-sampling, PSP pooling/concat, plane-sweep volume construction.
// All tensors are channels-last (NDHWC); every thread moves 16-byte chunks (8 bf16 / 4 f32).
#include "common.h"
#include "kernels.h"

namespace rgbm {

static inline unsigned grid_for(long long n, int block = 256) {
  long long g = (n + block - 1) / block;
  const long long cap = 256ll * 32;   // 256 CUs x 8 blocks, x4 oversubscription; grid-stride the rest
  return (unsigned)(g < cap ? (g > 0 ? g : 1) : cap);
}

// ---------------------------------------------------------------- NCHW fp32 -> NHWC(T), channel pad
template <typename T>
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ in, T* __restrict__ out, int V, int C, int H, int W,
                                        int Cp) {
  const long long total = (long long)V * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long v = i / ((long long)H * W);
    const long long hw = i - v * H * W;
    for (int c0 = 0; c0 < Cp; c0 += 4) {
      float vals[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        vals[e] = c < C ? in[(v * C + c) * H * W + hw] : 0.f;
      }
      store4(out + i * Cp + c0, vals);
    }
  }
}

int launch_nchw_to_nhwc_pad(int dtype, const float* in, void* out, int V, int C, int H, int W, int Cp, hipStream_t s) {
  const long long total = (long long)V * H * W;
  if (dtype == BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<unsigned short>, dim3(grid_for(total)), dim3(256), 0, s, in,
                       (unsigned short*)out, V, C, H, W, Cp);
  else if (dtype == F16)
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<f16_t>, dim3(grid_for(total)), dim3(256), 0, s, in,
                       (f16_t*)out, V, C, H, W, Cp);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<bx3_t>, dim3(grid_for(total)), dim3(256), 0, s, in,
                       (bx3_t*)out, V, C, H, W, Cp);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, in, (float*)out, V, C, H, W, Cp);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- max-pool 3x3 s2 p1 (pspnet.py:39)
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int V, int H, int W, int C, int Ho, int Wo) {
  constexpr int E = 16 / sizeof(T);
  const int cpp = C / E;
  const long long total = (long long)V * Ho * Wo * cpp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpp);
    long long t = i / cpp;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho); t /= Ho;
    const long long v = t;
    float m[E];
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = -INFINITY;
    for (int kh = 0; kh < 3; ++kh) {
      const int h = ho * 2 - 1 + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int w = wo * 2 - 1 + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        const uint4 c = *reinterpret_cast<const uint4*>(in + ((v * H + h) * W + w) * C + cc * E);
        float x[E];
        unpack_chunk(c, x, T());
#pragma unroll
        for (int e = 0; e < E; ++e) m[e] = (x[e] > m[e] || x[e] != x[e]) ? x[e] : m[e];
      }
    }
    *reinterpret_cast<uint4*>(out + ((v * Ho + ho) * Wo + wo) * C + cc * E) = pack_chunk(m, T());
  }
}

int launch_maxpool3x3s2(int dtype, const void* in, void* out, int V, int H, int W, int C, hipStream_t s) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int E = dtype_chunk(dtype);
  RGBM_REQUIRE(C % E == 0, "maxpool channels");
  const long long total = (long long)V * Ho * Wo * (C / E);
  if (dtype == BF16)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<unsigned short>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const unsigned short*)in, (unsigned short*)out, V, H, W, C, Ho, Wo);
  else if (dtype == F16)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<f16_t>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const f16_t*)in, (f16_t*)out, V, H, W, C, Ho, Wo);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<bx3_t>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const bx3_t*)in, (bx3_t*)out, V, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)in, (float*)out, V,
                       H, W, C, Ho, Wo);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- bilinear resize, align_corners=True
// (pspnet.py:93,106; torch upsample_bilinear2d: src = dst*(in-1)/(out-1), i1 = i0 + (i0 < in-1))
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_ac(int dst, int in_size, float scale) {
  Lerp l;
  // product and difference rounded separately, like torch's area_pixel_compute_source_index + lambda: left to the
  // optimiser, scale*dst - i0 becomes one fma in some instantiations and not in others (1-ulp weight differences
  // between the x2 kernel and the generic one)
  const float src = __fmul_rn(scale, (float)dst);
  l.i0 = (int)src;
  if (l.i0 > in_size - 1) l.i0 = in_size - 1;
  l.i1 = l.i0 + (l.i0 < in_size - 1 ? 1 : 0);
  l.w1 = __fsub_rn(src, (float)l.i0);
  l.w1 = l.w1 < 0.f ? 0.f : (l.w1 > 1.f ? 1.f : l.w1);
  l.w0 = 1.f - l.w1;
  return l;
}

// the one interpolation expression of both kernels, with its multiply-adds pinned (no compiler-chosen contraction)
__device__ __forceinline__ float bilerp(float a, float b, float c, float d, const Lerp& lx, const Lerp& ly) {
  const float top = fmaf(lx.w0, a, lx.w1 * b), bot = fmaf(lx.w0, c, lx.w1 * d);
  return fmaf(ly.w0, top, ly.w1 * bot);
}

template <typename T>
__global__ void resize_bilinear_ac_kernel(const T* __restrict__ in, T* __restrict__ out, int V, int Hs, int Ws, int C,
                                          int Ho, int Wo, int ldo, int ch_off, float sy, float sx) {
  constexpr int E = 16 / sizeof(T);
  const int cpp = C / E;
  const long long total = (long long)V * Ho * Wo * cpp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpp);
    long long t = i / cpp;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho); t /= Ho;
    const long long v = t;
    const Lerp ly = lerp_ac(ho, Hs, sy), lx = lerp_ac(wo, Ws, sx);
    const T* base = in + v * Hs * Ws * C + cc * E;
    float a[E], b[E], c[E], dd[E], r[E];
    unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)ly.i0 * Ws + lx.i0) * C), a, T());
    unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)ly.i0 * Ws + lx.i1) * C), b, T());
    unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)ly.i1 * Ws + lx.i0) * C), c, T());
    unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)ly.i1 * Ws + lx.i1) * C), dd, T());
#pragma unroll
    for (int e = 0; e < E; ++e)
      r[e] = bilerp(a[e], b[e], c[e], dd[e], lx, ly);
    *reinterpret_cast<uint4*>(out + ((v * Ho + ho) * Wo + wo) * ldo + ch_off + cc * E) = pack_chunk(r, T());
  }
}

template <typename T>
__device__ __forceinline__ void st_nt(T* p, const uint4& v) {          // streaming store: the output is far larger than L2 + MALL
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4*>(p));
}

// Exact x2 case (PSPUpsample, pspnet.py:100-107): one thread produces the 4 x 4 output block of the 2 x 2 input pixels
// (i..i+1, j..j+1).  In the interior its sixteen outputs tap rows i-1..i+2 and columns j-1..j+2: 16 chunk loads per 16
// outputs instead of 64 (the generic kernel is bound by L2 -> CU traffic, 4 x the output bytes; a 2 x 2 block per thread
// moved 2.25 x, 2 x 4 moved 1.5 x, this one 1 x).  Blocks whose taps differ (first row / column, float rounding at the last,
// odd height or width) take the generic 4-loads-per-output path.  Same weights and the same fused-multiply-add sequence
// (bilerp) as the generic kernel: identical results.
template <typename T>
__global__ __launch_bounds__(256) void resize2x_ac_kernel(const T* __restrict__ in, T* __restrict__ out, int V, int Hs, int Ws, int C,
                                                            int ldo, int ch_off, float sy, float sx) {
  constexpr int E = 16 / sizeof(T);
  const int cpp = C / E, Ho = 2 * Hs, Wo = 2 * Ws;
  const int Hp = (Hs + 1) >> 1, Wp = (Ws + 1) >> 1;              // 2 x 2 input blocks per column / row
  const long long total = (long long)V * Hp * Wp * cpp;
  // consecutive workgroup ids land on different XCDs (8, each with its own L2): give every XCD one contiguous eighth of the
  // 256-thread chunks, so that the image rows neighbouring chunks share are fetched into one L2 instead of up to three
  const long long nchunk = (total + 255) / 256, per_xcd = (nchunk + 7) / 8;
  for (long long q = blockIdx.x; q < per_xcd * 8; q += gridDim.x) {
    const long long chunk = (q & 7) * per_xcd + (q >> 3);
    const long long idx = chunk * 256 + threadIdx.x;
    if (chunk >= nchunk || idx >= total) continue;
    // 32-bit index arithmetic (the launcher checks total < 2^31): 64-bit divisions by run-time values cost ~100 instructions each
    const unsigned u = (unsigned)idx;
    const unsigned pp = u / (unsigned)cpp;
    const int cc = (int)(u - pp * (unsigned)cpp);
    const unsigned row = pp / (unsigned)Wp;
    const int j = 2 * (int)(pp - row * (unsigned)Wp);
    const unsigned vv = row / (unsigned)Hp;
    const int i = 2 * (int)(row - vv * (unsigned)Hp);
    const long long v = vv;
    const T* base = in + v * Hs * Ws * C + cc * E;
    T* obase = out + (v * Ho * Wo) * ldo + ch_off + cc * E;
    const int nrow = i + 1 < Hs ? 4 : 2, ncol = j + 1 < Ws ? 4 : 2;   // odd size: the last block holds one input row / column
    Lerp ly[4], lx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ly[k] = lerp_ac(min(2 * i + k, Ho - 1), Hs, sy);
      lx[k] = lerp_ac(min(2 * j + k, Wo - 1), Ws, sx);
    }
    bool fast = nrow == 4 && ncol == 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      fast = fast && ly[k].i0 == i - 1 + ((k + 1) >> 1) && ly[k].i1 == i + ((k + 1) >> 1) &&
             lx[k].i0 == j - 1 + ((k + 1) >> 1) && lx[k].i1 == j + ((k + 1) >> 1);
    if (fast) {
      uint4 p[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          p[a][b] = *reinterpret_cast<const uint4*>(base + ((long long)(i - 1 + a) * Ws + (j - 1 + b)) * C);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int r0 = (m + 1) >> 1;                                 // upper tap row of output row 2i+m, relative to i-1
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c0 = (k + 1) >> 1;                               // left tap column of output column 2j+k, relative to j-1
          float ta[E], tb[E], tc[E], td[E], r[E];
          unpack_chunk(p[r0][c0], ta, T());
          unpack_chunk(p[r0][c0 + 1], tb, T());
          unpack_chunk(p[r0 + 1][c0], tc, T());
          unpack_chunk(p[r0 + 1][c0 + 1], td, T());
#pragma unroll
          for (int e = 0; e < E; ++e) r[e] = bilerp(ta[e], tb[e], tc[e], td[e], lx[k], ly[m]);
          st_nt(obase + ((long long)(2 * i + m) * Wo + 2 * j + k) * ldo, pack_chunk(r, T()));
        }
      }
    } else {
      for (int m = 0; m < nrow; ++m) {
        const Lerp lym = lerp_ac(2 * i + m, Hs, sy);
        for (int k = 0; k < ncol; ++k) {
          const Lerp lxk = lerp_ac(2 * j + k, Ws, sx);
          float a[E], b[E], c[E], dd[E], r[E];
          unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)lym.i0 * Ws + lxk.i0) * C), a, T());
          unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)lym.i0 * Ws + lxk.i1) * C), b, T());
          unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)lym.i1 * Ws + lxk.i0) * C), c, T());
          unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)lym.i1 * Ws + lxk.i1) * C), dd, T());
#pragma unroll
          for (int e = 0; e < E; ++e) r[e] = bilerp(a[e], b[e], c[e], dd[e], lxk, lym);
          st_nt(obase + ((long long)(2 * i + m) * Wo + 2 * j + k) * ldo, pack_chunk(r, T()));
        }
      }
    }
  }
}

int launch_resize_bilinear_ac(int dtype, const void* in, void* out, int V, int Hs, int Ws, int C, int Ho, int Wo, int ldo,
                              int ch_off, hipStream_t s) {
  const int E = dtype_chunk(dtype);
  RGBM_REQUIRE(C % E == 0 && ldo % E == 0 && ch_off % E == 0, "resize channel alignment");
  const float sy = Ho > 1 ? (float)(Hs - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(Ws - 1) / (float)(Wo - 1) : 0.f;
  if (Ho == 2 * Hs && Wo == 2 * Ws && Hs >= 2 && Ws >= 2 && (long long)V * Hs * Ws * (C / E) < (1ll << 31) && !(g_debug_flags & 512)) {
    const long long blocks = (long long)V * ((Hs + 1) / 2) * ((Ws + 1) / 2) * (C / E);      // one thread per 2 x 2 input block and 16-byte chunk
    if (dtype == BF16)
      hipLaunchKernelGGL(resize2x_ac_kernel<unsigned short>, dim3(grid_for(blocks)), dim3(256), 0, s, (const unsigned short*)in,
                         (unsigned short*)out, V, Hs, Ws, C, ldo, ch_off, sy, sx);
    else if (dtype == F16)
      hipLaunchKernelGGL(resize2x_ac_kernel<f16_t>, dim3(grid_for(blocks)), dim3(256), 0, s, (const f16_t*)in,
                         (f16_t*)out, V, Hs, Ws, C, ldo, ch_off, sy, sx);
    else if (dtype == BF16X3)
      hipLaunchKernelGGL(resize2x_ac_kernel<bx3_t>, dim3(grid_for(blocks)), dim3(256), 0, s, (const bx3_t*)in,
                         (bx3_t*)out, V, Hs, Ws, C, ldo, ch_off, sy, sx);
    else
      hipLaunchKernelGGL(resize2x_ac_kernel<float>, dim3(grid_for(blocks)), dim3(256), 0, s, (const float*)in, (float*)out, V, Hs,
                         Ws, C, ldo, ch_off, sy, sx);
    RGBM_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const long long total = (long long)V * Ho * Wo * (C / E);
  if (dtype == BF16)
    hipLaunchKernelGGL(resize_bilinear_ac_kernel<unsigned short>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const unsigned short*)in, (unsigned short*)out, V, Hs, Ws, C, Ho, Wo, ldo, ch_off, sy, sx);
  else if (dtype == F16)
    hipLaunchKernelGGL(resize_bilinear_ac_kernel<f16_t>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const f16_t*)in, (f16_t*)out, V, Hs, Ws, C, Ho, Wo, ldo, ch_off, sy, sx);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(resize_bilinear_ac_kernel<bx3_t>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const bx3_t*)in, (bx3_t*)out, V, Hs, Ws, C, Ho, Wo, ldo, ch_off, sy, sx);
  else
    hipLaunchKernelGGL(resize_bilinear_ac_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)in,
                       (float*)out, V, Hs, Ws, C, Ho, Wo, ldo, ch_off, sy, sx);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- channel-slice copy (concat helper)
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ in, T* __restrict__ out, long long npix, int C, int ldo, int ch_off) {
  constexpr int E = 16 / sizeof(T);
  const int cpp = C / E;
  const long long total = npix * cpp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpp);
    const long long p = i / cpp;
    *reinterpret_cast<uint4*>(out + p * ldo + ch_off + cc * E) = *reinterpret_cast<const uint4*>(in + p * C + cc * E);
  }
}

int launch_copy_channels(int dtype, const void* in, void* out, long long npix, int C, int ldo, int ch_off, hipStream_t s) {
  const int E = dtype_chunk(dtype);
  RGBM_REQUIRE(C % E == 0 && ldo % E == 0 && ch_off % E == 0, "copy channel alignment");
  const long long total = npix * (C / E);
  if (dtype == BF16)
    hipLaunchKernelGGL(copy_channels_kernel<unsigned short>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const unsigned short*)in, (unsigned short*)out, npix, C, ldo, ch_off);
  else if (dtype == F16)
    hipLaunchKernelGGL(copy_channels_kernel<f16_t>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const f16_t*)in, (f16_t*)out, npix, C, ldo, ch_off);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(copy_channels_kernel<bx3_t>, dim3(grid_for(total)), dim3(256), 0, s,
                       (const bx3_t*)in, (bx3_t*)out, npix, C, ldo, ch_off);
  else
    hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)in, (float*)out,
                       npix, C, ldo, ch_off);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- adaptive average pool (pspnet.py:83)
// window of output cell i over an axis of length L with S bins: [floor(i*L/S), ceil((i+1)*L/S))
// One launch pools every bin size of the pyramid (pspnet.py:83-94: bins 1, 2, 3, 6): block = (view, cell of one of the
// grids), 8 pixel lanes x C/E channel chunks; the window's pixels are split over the pixel lanes and reduced through LDS.
// (Four launches with one 64-thread block per cell walked up to 784 pixels serially: 0.17 ms each, 512 blocks for S = 1.)
struct PoolBins { void* out[4]; int S[4]; int first[5]; int n; };      // first[i]: index of grid i's first cell, first[n] = total

template <typename T>
__global__ __launch_bounds__(512) void adaptive_avgpool_kernel(const T* __restrict__ in, const PoolBins pb, int V, int H, int W, int C) {
  constexpr int E = 16 / sizeof(T);
  constexpr int PL = 8;                                   // pixel lanes
  __shared__ float red[PL][64][E];
  const int ncell = pb.first[pb.n];
  const int cell_all = blockIdx.x % ncell;
  const long long v = blockIdx.x / ncell;
  int g = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i < pb.n && cell_all >= pb.first[i]) g = i;
  const int S = pb.S[g], cell = cell_all - pb.first[g];
  const int sy = cell / S, sx = cell % S;
  const int h0 = (sy * H) / S, h1 = ((sy + 1) * H + S - 1) / S;
  const int w0 = (sx * W) / S, w1 = ((sx + 1) * W + S - 1) / S;
  const int ww = w1 - w0, npx = (h1 - h0) * ww;
  const float inv = 1.0f / (float)npx;
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  T* __restrict__ out = reinterpret_cast<T*>(pb.out[g]);
  for (int c0 = 0; c0 < C / E; c0 += 64) {
    const int cc = c0 + cl;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    if (cc < C / E)
#pragma unroll 4                                            // four loads in flight per thread (round 5: the one-bin cell walks 98 pixels per lane; 39 -> 14 us at B = 1); same order of the sums
      for (int p = pl; p < npx; p += PL) {
        const int h = h0 + p / ww, w = w0 + p % ww;
        float x[E];
        unpack_chunk(*reinterpret_cast<const uint4*>(in + ((v * H + h) * W + w) * C + cc * E), x, T());
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += x[e];
      }
#pragma unroll
    for (int e = 0; e < E; ++e) red[pl][cl][e] = acc[e];
    __syncthreads();
    if (pl == 0 && cc < C / E) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float t = red[0][cl][e];
#pragma unroll
        for (int q = 1; q < PL; ++q) t += red[q][cl][e];
        acc[e] = t * inv;
      }
      *reinterpret_cast<uint4*>(out + ((long long)v * S * S + cell) * C + cc * E) = pack_chunk(acc, T());
    }
    __syncthreads();
  }
}

int launch_adaptive_avgpool_multi(int dtype, const void* in, void* const* outs, const int* bins, int nb, int V, int H, int W, int C,
                                  hipStream_t s) {
  const int E = dtype_chunk(dtype);
  RGBM_REQUIRE(C % E == 0 && nb >= 1 && nb <= 4, "avgpool channel alignment / bin count");
  PoolBins pb;
  pb.n = nb;
  pb.first[0] = 0;
  for (int i = 0; i < 4; ++i) {
    pb.out[i] = i < nb ? outs[i] : nullptr;
    pb.S[i] = i < nb ? bins[i] : 1;
    pb.first[i + 1] = pb.first[i] + (i < nb ? bins[i] * bins[i] : 0);
  }
  pb.first[nb] = pb.first[nb];
  const long long blocks = (long long)V * pb.first[nb];
  RGBM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "avgpool grid");
  if (dtype == BF16)
    hipLaunchKernelGGL(adaptive_avgpool_kernel<unsigned short>, dim3((unsigned)blocks), dim3(512), 0, s, (const unsigned short*)in, pb, V, H, W, C);
  else if (dtype == F16)
    hipLaunchKernelGGL(adaptive_avgpool_kernel<f16_t>, dim3((unsigned)blocks), dim3(512), 0, s, (const f16_t*)in, pb, V, H, W, C);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(adaptive_avgpool_kernel<bx3_t>, dim3((unsigned)blocks), dim3(512), 0, s, (const bx3_t*)in, pb, V, H, W, C);
  else
    hipLaunchKernelGGL(adaptive_avgpool_kernel<float>, dim3((unsigned)blocks), dim3(512), 0, s, (const float*)in, pb, V, H, W, C);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_adaptive_avgpool(int dtype, const void* in, void* out, int V, int H, int W, int C, int S, hipStream_t s) {
  void* outs[1] = {out};
  return launch_adaptive_avgpool_multi(dtype, in, outs, &S, 1, V, H, W, C, s);
}

// ---------------------------------------------------------------- PSP stage for small batches (pspnet.py:76-94)
// The pyramid of a deployment-sized batch (B = 1 .. 8: 2 .. 16 views) is 50 cells per view, each a 512 -> 128 1x1 conv of one pooled
// vector: four GEMM launches of 2 .. 576 rows behind the pooling launch, and four resize launches behind them — ten launches of
// 5-25 us for a few MFLOP.  psp_pool_conv_kernel: one workgroup per (view, cell) pools its window exactly as adaptive_avgpool_kernel
// does (same split over the pixel lanes, same order of the sums, the mean rounded to the storage type as the pooled tensor was) and
// multiplies it with the stage's [128][512] weights on the vector pipe: a wave owns 16 output channels, a lane one 16-byte chunk of
// K (two for the 4-byte types), fp32 FMAs, butterfly sum over the lanes — a fixed order.  psp_resize_cat_kernel: the four stages'
// bilinear resize (align_corners, the arithmetic of resize_bilinear_ac_kernel) into their channel slices of `cat` together with the
// copy of the backbone's 512 channels: the whole concat in one launch, at any batch size.
struct PspStage { const void* in[4]; const void* w[4]; void* out[4]; int S[4]; int first[5]; float sy[4], sx[4]; };

template <typename T>
__global__ __launch_bounds__(512) void psp_pool_conv_kernel(const T* __restrict__ in, const PspStage ps, int V, int H, int W, int ldi, int act,
                                                              float slope) {
  constexpr int E = 16 / sizeof(T);
  constexpr int C = 512, CO = 128, NCH = C / E;            // 64 chunks (16-bit types) or 128
  constexpr int PL = 8;
  __shared__ float red[PL][64][E];
  __shared__ uint4 pooled[NCH];
  __shared__ float res[CO];
  const int ncell = ps.first[4];
  const int cell_all = blockIdx.x % ncell;
  const long long v = blockIdx.x / ncell;
  int g = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (cell_all >= ps.first[i]) g = i;
  const int S = ps.S[g], cell = cell_all - ps.first[g];
  const int sy = cell / S, sx = cell % S;
  const int h0 = (sy * H) / S, h1 = ((sy + 1) * H + S - 1) / S;
  const int w0 = (sx * W) / S, w1 = ((sx + 1) * W + S - 1) / S;
  const int ww = w1 - w0, npx = (h1 - h0) * ww;
  const float inv = 1.0f / (float)npx;
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  for (int c0 = 0; c0 < NCH; c0 += 64) {
    const int cc = c0 + cl;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll 8                                            // (eight loads in flight per thread: the one-bin cell walks 98 pixels per lane; same order of the sums)
    for (int p = pl; p < npx; p += PL) {
      const int h = h0 + p / ww, w = w0 + p % ww;
      float x[E];
      unpack_chunk(*reinterpret_cast<const uint4*>(in + ((v * H + h) * W + w) * ldi + cc * E), x, T());
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] += x[e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) red[pl][cl][e] = acc[e];
    __syncthreads();
    if (pl == 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float t = red[0][cl][e];
#pragma unroll
        for (int q = 1; q < PL; ++q) t += red[q][cl][e];
        acc[e] = t * inv;
      }
      pooled[cc] = pack_chunk(acc, T());
    }
    __syncthreads();
  }
  // 1x1 conv: wave `pl` owns output channels 16 pl .. 16 pl + 15
  const T* __restrict__ wg = reinterpret_cast<const T*>(ps.w[g]);
  float x[NCH / 64][E];
#pragma unroll
  for (int q = 0; q < NCH / 64; ++q) unpack_chunk(pooled[q * 64 + cl], x[q], T());
  float part[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NCH / 64; ++q) {
      float wv[E];
      unpack_chunk(*reinterpret_cast<const uint4*>(wg + (long long)(pl * 16 + o) * C + (q * 64 + cl) * E), wv, T());
#pragma unroll
      for (int e = 0; e < E; ++e) sum = fmaf(wv[e], x[q][e], sum);
    }
    part[o] = sum;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int o = 0; o < 16; ++o) part[o] += __shfl_xor(part[o], m, 64);
  if (cl == 0) {
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const float r = part[o];
      res[pl * 16 + o] = act == ACT_RELU ? (r < 0.f ? 0.f : r) : act == ACT_PRELU ? (r < 0.f ? r * slope : r) : r;
    }
  }
  __syncthreads();
  if (threadIdx.x < CO / E) {
    float r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = res[threadIdx.x * E + e];
    T* __restrict__ out = reinterpret_cast<T*>(ps.out[g]);
    *reinterpret_cast<uint4*>(out + ((long long)v * S * S + cell) * CO + threadIdx.x * E) = pack_chunk(r, T());
  }
}

template <typename T>
__global__ __launch_bounds__(256) void psp_resize_cat_kernel(const T* __restrict__ f, const PspStage ps, T* __restrict__ out, int V, int Ho, int Wo) {
  constexpr int E = 16 / sizeof(T);
  constexpr int CF = 512, CO = 128, CPS = CO / E, CPF = CF / E, CPP = CPF + 4 * CPS;      // chunks per stage / of the backbone's channels / per pixel of `out`
  const long long total = (long long)V * Ho * Wo * CPP;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cp = (int)(i % CPP);
    long long t = i / CPP;
    if (cp < CPF) {                                     // channels 0 .. 511: the backbone's feature map itself (pspnet.py:93 `[feats]`)
      reinterpret_cast<uint4*>(out)[i] = *reinterpret_cast<const uint4*>(f + t * CF + cp * E);
      continue;
    }
    const int cq = cp - CPF;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho); t /= Ho;
    const long long v = t;
    const int g = cq / CPS, cc = cq - g * CPS;
    const int S = ps.S[g];
    const Lerp ly = lerp_ac(ho, S, ps.sy[g]), lx = lerp_ac(wo, S, ps.sx[g]);
    const T* base = reinterpret_cast<const T*>(ps.in[g]) + v * S * S * CO + cc * E;
    float a[E], b[E], c[E], dd[E], r[E];
    unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)ly.i0 * S + lx.i0) * CO), a, T());
    unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)ly.i0 * S + lx.i1) * CO), b, T());
    unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)ly.i1 * S + lx.i0) * CO), c, T());
    unpack_chunk(*reinterpret_cast<const uint4*>(base + ((long long)ly.i1 * S + lx.i1) * CO), dd, T());
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = bilerp(a[e], b[e], c[e], dd[e], lx, ly);
    reinterpret_cast<uint4*>(out)[i] = pack_chunk(r, T());      // out is [V][Ho][Wo][1024]: chunk i of the tensor
  }
}

static int psp_stage_desc(const int* bins, PspStage& ps) {
  ps.first[0] = 0;
  for (int i = 0; i < 4; ++i) {
    RGBM_REQUIRE(bins[i] >= 1, "psp bin size");
    ps.S[i] = bins[i];
    ps.first[i + 1] = ps.first[i] + bins[i] * bins[i];
  }
  return 0;
}

// pooled cells of the four bin sizes x their 512 -> 128 1x1 convs (weights [128][512] in the storage type, no bias) -> outs[i] [V][S_i][S_i][128]
int launch_psp_pool_conv(int dtype, const void* in, int ldi, const void* const* w, void* const* outs, const int* bins, int V, int H, int W,
                         int act, float slope, hipStream_t s) {
  const int E = dtype_chunk(dtype);
  RGBM_REQUIRE(ldi % E == 0 && ldi >= 512 && V > 0, "psp stage arguments");
  PspStage ps{};
  if (int rc = psp_stage_desc(bins, ps)) return rc;
  for (int i = 0; i < 4; ++i) { ps.w[i] = w[i]; ps.out[i] = outs[i]; }
  const long long blocks = (long long)V * ps.first[4];
  RGBM_REQUIRE(blocks < (1ll << 31), "psp stage grid");
  if (dtype == BF16)
    hipLaunchKernelGGL(psp_pool_conv_kernel<unsigned short>, dim3((unsigned)blocks), dim3(512), 0, s, (const unsigned short*)in, ps, V, H, W, ldi, act, slope);
  else if (dtype == F16)
    hipLaunchKernelGGL(psp_pool_conv_kernel<f16_t>, dim3((unsigned)blocks), dim3(512), 0, s, (const f16_t*)in, ps, V, H, W, ldi, act, slope);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(psp_pool_conv_kernel<bx3_t>, dim3((unsigned)blocks), dim3(512), 0, s, (const bx3_t*)in, ps, V, H, W, ldi, act, slope);
  else
    hipLaunchKernelGGL(psp_pool_conv_kernel<float>, dim3((unsigned)blocks), dim3(512), 0, s, (const float*)in, ps, V, H, W, ldi, act, slope);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// out [V][Ho][Wo][1024] = concat(f [V][Ho][Wo][512], the four stages [V][S_i][S_i][128] resized (bilinear, align_corners) to Ho x Wo)
int launch_psp_resize_cat(int dtype, const void* f, const void* const* stages, const int* bins, void* out, int V, int Ho, int Wo, hipStream_t s) {
  const int E = dtype_chunk(dtype);
  RGBM_REQUIRE(V > 0 && f && out, "psp resize arguments");
  PspStage ps{};
  if (int rc = psp_stage_desc(bins, ps)) return rc;
  for (int i = 0; i < 4; ++i) {
    ps.in[i] = stages[i];
    ps.sy[i] = Ho > 1 ? (float)(bins[i] - 1) / (float)(Ho - 1) : 0.f;
    ps.sx[i] = Wo > 1 ? (float)(bins[i] - 1) / (float)(Wo - 1) : 0.f;
  }
  const long long total = (long long)V * Ho * Wo * (1024 / E);
  if (dtype == BF16)
    hipLaunchKernelGGL(psp_resize_cat_kernel<unsigned short>, dim3(grid_for(total)), dim3(256), 0, s, (const unsigned short*)f, ps, (unsigned short*)out, V, Ho, Wo);
  else if (dtype == F16)
    hipLaunchKernelGGL(psp_resize_cat_kernel<f16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const f16_t*)f, ps, (f16_t*)out, V, Ho, Wo);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(psp_resize_cat_kernel<bx3_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bx3_t*)f, ps, (bx3_t*)out, V, Ho, Wo);
  else
    hipLaunchKernelGGL(psp_resize_cat_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)f, ps, (float*)out, V, Ho, Wo);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- plane-sweep homography (network_v5.py:390-392)
// proj = P_src @ inverse(P_ref).  The 4x4 inverse is done in fp64 (Gauss-Jordan, partial pivoting) and
// rounded to fp32; the product is accumulated in fp32 like torch.matmul.  out[v] = {rot[9] row-major, trans[3]}.
__global__ void homography_kernel(const float* __restrict__ P, float* __restrict__ out, int V, int B) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const int partner = (v + B) % V;
  const float* Pref = P + (long long)v * 16;
  const float* Psrc = P + (long long)partner * 16;
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int jj = 0; jj < 4; ++jj) { a[i][jj] = (double)Pref[i * 4 + jj]; a[i][4 + jj] = i == jj ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {
    int piv = c; double best = fabs(a[c][c]);
    for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
    if (piv != c) for (int k = 0; k < 8; ++k) { double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
    const double inv = 1.0 / a[c][c];     // singular -> inf/nan, propagates like torch.inverse garbage -> default bbox
    for (int k = 0; k < 8; ++k) a[c][k] *= inv;
    for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k]; }
  }
  float inv32[16];
  for (int i = 0; i < 4; ++i) for (int jj = 0; jj < 4; ++jj) inv32[i * 4 + jj] = (float)a[i][4 + jj];
  float* o = out + (long long)v * 12;
  for (int i = 0; i < 3; ++i)
    for (int jj = 0; jj < 4; ++jj) {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc = fmaf(Psrc[i * 4 + k], inv32[k * 4 + jj], acc);
      if (jj < 3) o[i * 3 + jj] = acc; else o[9 + i] = acc;
    }
}

int launch_homography(const float* P_views, float* out, int V, int B, hipStream_t s) {
  hipLaunchKernelGGL(homography_kernel, dim3((V + 63) / 64), dim3(64), 0, s, P_views, out, V, B);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- fused plane-sweep volume (network_v5.py:378-430)
// vol[v,d,y,x,:] = feat[v,y,x,:] + grid_sample(feat[partner(v)], homography(v,d,y,x))   (bilinear, zeros,
// align_corners=False fed with the align_corners=True normalisation, exactly as the reference does).
__device__ __forceinline__ void warp_coords(const float* __restrict__ hm, float x, float y, float depth, int H, int W,
                                            float& ix, float& iy) {
  // rot_xyz = rot @ [x,y,1]; * depth; + trans; perspective divide; normalise; un-normalise (grid_sample)
  const float rx = hm[0] * x + hm[1] * y + hm[2];
  const float ry = hm[3] * x + hm[4] * y + hm[5];
  const float rz = hm[6] * x + hm[7] * y + hm[8];
  const float px = rx * depth + hm[9], py = ry * depth + hm[10], pz = rz * depth + hm[11];
  const float u = px / pz, vv = py / pz;
  const float gx = u / ((float)(W - 1) / 2.f) - 1.f;
  const float gy = vv / ((float)(H - 1) / 2.f) - 1.f;
  ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
  iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
}

struct Bilin { int x0, y0; float w00, w01, w10, w11; bool nan; };
__device__ __forceinline__ Bilin bilin_setup(float ix, float iy) {
  Bilin b;
  b.nan = !(isfinite(ix) && isfinite(iy));
  if (b.nan) { b.x0 = b.y0 = -100; b.w00 = b.w01 = b.w10 = b.w11 = 0.f; return b; }
  // clamp far-out-of-range coordinates before the int conversion (all four taps are outside anyway)
  ix = fminf(fmaxf(ix, -4.f), 1.0e6f);
  iy = fminf(fmaxf(iy, -4.f), 1.0e6f);
  const float fx = floorf(ix), fy = floorf(iy);
  b.x0 = (int)fx; b.y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  b.w00 = (1.f - tx) * (1.f - ty);   // (x0,y0)  "nw"
  b.w01 = tx * (1.f - ty);           // (x1,y0)  "ne"
  b.w10 = (1.f - tx) * ty;           // (x0,y1)  "sw"
  b.w11 = tx * ty;                   // (x1,y1)  "se"
  return b;
}

template <typename T>
__global__ void build_volume_kernel(const T* __restrict__ feat, const float* __restrict__ homog, const float* __restrict__ depths,
                                    T* __restrict__ vol, int v0, int Vc, int V, int B, int D, int H, int W) {
  constexpr int C = 32;
  constexpr int E = 16 / sizeof(T);
  constexpr int NCH = C / E;
  const long long total = (long long)Vc * D * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long t = i;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); t /= H;
    const int dz = (int)(t % D); t /= D;
    const int v = v0 + (int)t;
    const int partner = (v + B) % V;
    const int b = v % B;
    float ix, iy;
    warp_coords(homog + (long long)v * 12, (float)x, (float)y, depths[b * D + dz], H, W, ix, iy);
    const Bilin bl = bilin_setup(ix, iy);
    const T* ref = feat + (((long long)v * H + y) * W + x) * C;
    const T* src = feat + (long long)partner * H * W * C;
    const bool in00 = (unsigned)bl.x0 < (unsigned)W && (unsigned)bl.y0 < (unsigned)H;
    const bool in01 = (unsigned)(bl.x0 + 1) < (unsigned)W && (unsigned)bl.y0 < (unsigned)H;
    const bool in10 = (unsigned)bl.x0 < (unsigned)W && (unsigned)(bl.y0 + 1) < (unsigned)H;
    const bool in11 = (unsigned)(bl.x0 + 1) < (unsigned)W && (unsigned)(bl.y0 + 1) < (unsigned)H;
    T* o = vol + i * C;
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      float r[E], acc[E];
      unpack_chunk(*reinterpret_cast<const uint4*>(ref + cc * E), r, T());
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = 0.f;
      float s[E];
      if (in00) { unpack_chunk(*reinterpret_cast<const uint4*>(src + ((long long)bl.y0 * W + bl.x0) * C + cc * E), s, T());
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += s[e] * bl.w00; }
      if (in01) { unpack_chunk(*reinterpret_cast<const uint4*>(src + ((long long)bl.y0 * W + bl.x0 + 1) * C + cc * E), s, T());
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += s[e] * bl.w01; }
      if (in10) { unpack_chunk(*reinterpret_cast<const uint4*>(src + ((long long)(bl.y0 + 1) * W + bl.x0) * C + cc * E), s, T());
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += s[e] * bl.w10; }
      if (in11) { unpack_chunk(*reinterpret_cast<const uint4*>(src + ((long long)(bl.y0 + 1) * W + bl.x0 + 1) * C + cc * E), s, T());
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += s[e] * bl.w11; }
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = bl.nan ? __builtin_nanf("") : (r[e] + acc[e]);
      *reinterpret_cast<uint4*>(o + cc * E) = pack_chunk(acc, T());
    }
  }
}

int launch_build_volume(int dtype, const void* feat, const float* homog, const float* depths, void* vol, int v0, int Vc, int V,
                        int B, int D, int H, int W, hipStream_t s) {
  const long long total = (long long)Vc * D * H * W;
  if (dtype == BF16)
    hipLaunchKernelGGL(build_volume_kernel<unsigned short>, dim3(grid_for(total) * 4), dim3(256), 0, s,
                       (const unsigned short*)feat, homog, depths, (unsigned short*)vol, v0, Vc, V, B, D, H, W);
  else if (dtype == F16)
    hipLaunchKernelGGL(build_volume_kernel<f16_t>, dim3(grid_for(total) * 4), dim3(256), 0, s,
                       (const f16_t*)feat, homog, depths, (f16_t*)vol, v0, Vc, V, B, D, H, W);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(build_volume_kernel<bx3_t>, dim3(grid_for(total) * 4), dim3(256), 0, s,
                       (const bx3_t*)feat, homog, depths, (bx3_t*)vol, v0, Vc, V, B, D, H, W);
  else
    hipLaunchKernelGGL(build_volume_kernel<float>, dim3(grid_for(total) * 4), dim3(256), 0, s, (const float*)feat, homog,
                       depths, (float*)vol, v0, Vc, V, B, D, H, W);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- dtype -> fp32 copy (debug / test fetch)
template <typename T>
__global__ void to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = Elem<T>::ld(in + i);
}

int launch_to_f32(int dtype, const void* in, float* out, long long n, hipStream_t s) {
  if (dtype == BF16)
    hipLaunchKernelGGL(to_f32_kernel<unsigned short>, dim3(grid_for(n)), dim3(256), 0, s, (const unsigned short*)in, out, n);
  else if (dtype == F16)
    hipLaunchKernelGGL(to_f32_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, s, (const f16_t*)in, out, n);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(to_f32_kernel<bx3_t>, dim3(grid_for(n)), dim3(256), 0, s, (const bx3_t*)in, out, n);
  else
    hipLaunchKernelGGL(to_f32_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (const float*)in, out, n);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- split-pair (BF16X3) -> plain fp32, one 16-byte chunk per thread
__global__ __launch_bounds__(256) void bx3_to_f32_kernel(const uint4* __restrict__ in, float4* __restrict__ out, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float v[4];
    bx3_join4(in[i], v);
    out[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

int launch_bx3_to_f32(const void* in, float* out, long long n, hipStream_t s) {
  RGBM_REQUIRE(n % 4 == 0, "bx3_to_f32: element count must be a multiple of 4");
  hipLaunchKernelGGL(bx3_to_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, (const uint4*)in, (float4*)out, n / 4);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- fp32 -> split pairs (input of the split-pair pose MLP); in place is allowed
__global__ __launch_bounds__(256) void f32_to_bx3_kernel(const float4* in, uint4* out, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 a = in[i];
    out[i] = bx3_split4(a.x, a.y, a.z, a.w);
  }
}

int launch_f32_to_bx3(const float* in, void* out, long long n, hipStream_t s) {
  RGBM_REQUIRE(n % 4 == 0, "f32_to_bx3: element count must be a multiple of 4");
  hipLaunchKernelGGL(f32_to_bx3_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, (const float4*)in, (uint4*)out, n / 4);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- fp32 -> fp16 copy (input of the fp16 pose MLP), 8 elements per thread
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ in, f16_t* __restrict__ out, long long n8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(in + i * 8), b = *reinterpret_cast<const float4*>(in + i * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *reinterpret_cast<uint4*>(out + i * 8) = pack_chunk(v, f16_t());
  }
}

int launch_f32_to_f16(const float* in, void* out, long long n, hipStream_t s) {
  RGBM_REQUIRE(n % 8 == 0, "f32_to_f16: element count must be a multiple of 8");
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, in, (f16_t*)out, n / 8);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
