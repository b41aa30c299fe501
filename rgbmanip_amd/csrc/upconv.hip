// PSPUpsample (pspnet.py:100-107: x2 bilinear, align_corners=True -> conv3x3 pad 1 + bias -> PReLU) without the up-sampled tensor.
//
// The bilinear up-sampling U acts on space only, a tap's weight matrix W_t on channels only, so they commute:
//     conv3x3(U x)(p) = sum_t W_t . (U x)(p + t) = sum_t (U (W_t . x))(p + t)            (zero outside the up-sampled grid)
// Step 1 is therefore a 1x1 convolution at the LOW resolution with the nine taps stacked on the output channels
// (z[v][i][j][t*Co + co] = sum_ci W[co][ci][t] x[v][i][j][ci]: a quarter of the multiply-adds of the 3x3 conv on the
// up-sampled grid, run by the implicit-GEMM kernels), step 2 this kernel: every output pixel gathers, per tap, the bilinear
// interpolation of that tap's z plane at the tap-shifted position and adds the nine results, bias and activation.
// The 4x larger up-sampled input (3.3 GB per launch at batch 256 in bf16) is never written or read.
//
// Interpolation: output position u of an x2 align_corners grid reads source position u*(n-1)/(2n-1), which lies in
// [I-1, I] for u = 2I and in [I, I+1/2] for u = 2I+1.  A thread owns a BR x BC output block; the BR+2 rows / BC+2 columns its
// taps reach use low-resolution rows base + (k >> 1) and the one after it, so the weights are formed relative to that fixed
// pattern (w1 = src - row, w0 = 1 - w1) instead of through floor(): the piecewise-linear interpolant is continuous, so a
// source position that float rounding puts 1e-7 on the other side of a knot gives the same value to rounding, and every
// thread runs the same straight-line code.  Rows / columns outside the low-resolution image are clamped for the load (their
// weight is 0 or O(1e-7)); positions outside the up-sampled grid (the conv's zero padding) get weight 0.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace rgbm {

namespace {

// A thread handles 4 channels of its block positions in every storage type: 8 bytes per load for the 16-bit types, 16 for f32 /
// split pairs (with 8 channels per thread the two register sets of z chunks, the block's accumulators and a row of interpolated
// values no longer fit 256 registers)
template <typename T> struct Raw4 { typedef uint4 type; };
template <> struct Raw4<unsigned short> { typedef uint2 type; };
template <> struct Raw4<f16_t> { typedef uint2 type; };
__device__ __forceinline__ void unpack4(const uint2& r, float* v, unsigned short) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void unpack4(const uint2& r, float* v, f16_t) {
  const f16x4 hh = __builtin_bit_cast(f16x4, r);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (float)hh[e];
}
__device__ __forceinline__ void unpack4(const uint4& r, float* v, float) { unpack_chunk(r, v, float()); }
__device__ __forceinline__ void unpack4(const uint4& r, float* v, bx3_t) { bx3_join4(r, v); }

// One tap of a thread's block.  raw: the tap's z chunks at the block's NR x NC low-resolution pixels (only the rows / columns the
// tap's output rows / columns touch are loaded and read).
template <typename T, int BR, int BC, int KH, int KW>
struct UpTap {
  static constexpr int E = 4;
  typedef typename Raw4<T>::type raw_t;
  static constexpr int NR = BR / 2 + 2, NC = BC / 2 + 2;
  static constexpr int JMIN = KH >> 1, JMAX = ((KH + BR - 1) >> 1) + 1;
  static constexpr int CMIN = KW >> 1, CMAX = ((KW + BC - 1) >> 1) + 1;
  __device__ static __forceinline__ void load(const T* zb, int Co, long long ldz, const int (&roff)[NR], const int (&coff)[NC],
                                              raw_t (&raw)[NR][NC]) {
    const T* zt = zb + (KH * 3 + KW) * Co;
#pragma unroll
    for (int j = JMIN; j <= JMAX; ++j)
#pragma unroll
      for (int c = CMIN; c <= CMAX; ++c) raw[j][c] = *reinterpret_cast<const raw_t*>(zt + (long long)(roff[j] + coff[c]) * ldz);
  }
  __device__ static __forceinline__ void accumulate(const raw_t (&raw)[NR][NC], const float (&wy)[BR + 2][2],
                                                    const float (&wx)[BC + 2][2], float (&acc)[BR][BC][E]) {
#pragma unroll
    for (int j = JMIN; j <= JMAX; ++j) {
      float hx[BC][E];
#pragma unroll
      for (int b = 0; b < BC; ++b) {
        const int c0 = (b + KW) >> 1;
        float p0[E], p1[E];
        unpack4(raw[j][c0], p0, T());
        unpack4(raw[j][c0 + 1], p1, T());
#pragma unroll
        for (int e = 0; e < E; ++e) hx[b][e] = fmaf(wx[b + KW][0], p0[e], wx[b + KW][1] * p1[e]);
      }
#pragma unroll
      for (int a = 0; a < BR; ++a) {
        const int ra = (a + KH) >> 1;
        if (j != ra && j != ra + 1) continue;
        const float coef = wy[a + KH][j == ra ? 0 : 1];
#pragma unroll
        for (int b = 0; b < BC; ++b)
#pragma unroll
          for (int e = 0; e < E; ++e) acc[a][b][e] = fmaf(coef, hx[b][e], acc[a][b][e]);
      }
      // pin the accumulators here: hipcc otherwise places every accumulation chain next to its only use, the final stores,
      // and carries all interpolated values of all nine taps until then (the scheduling barriers order machine instructions,
      // not where instruction selection first puts pure arithmetic)
#pragma unroll
      for (int a = 0; a < BR; ++a)
#pragma unroll
        for (int b = 0; b < BC; ++b)
          asm volatile("" : "+v"(acc[a][b][0]), "+v"(acc[a][b][1]), "+v"(acc[a][b][2]), "+v"(acc[a][b][3]));
      __builtin_amdgcn_sched_barrier(0);      // row by row: all rows' interpolated values at once cost another 2 x BC x E registers
    }
  }
};

template <typename T, int BR, int BC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? 3 : 2, 8))) void upconv_combine_kernel(const T* __restrict__ z, const float* __restrict__ bias,
                                                               T* __restrict__ out, int V, int h, int w, int Co, int ldo,
                                                               float nslope, float sy, float sx) {
  constexpr int E = 4;
  typedef typename Raw4<T>::type raw_t;
  constexpr int NR = BR / 2 + 2, NC = BC / 2 + 2;
  const int cpp = Co / E, Ho = 2 * h, Wo = 2 * w;
  const int nbi = (Ho + BR - 1) / BR, nbj = (Wo + BC - 1) / BC;
  const long long total = (long long)V * nbi * nbj * cpp;
  const long long ldz = 9ll * Co;
  // every XCD (own L2) takes one contiguous eighth of the 256-thread chunks: neighbouring blocks share low-resolution rows
  const long long nchunk = (total + 255) / 256, per_xcd = (nchunk + 7) / 8;
  for (long long q = blockIdx.x; q < per_xcd * 8; q += gridDim.x) {
    const long long chunk = (q & 7) * per_xcd + (q >> 3);
    const long long idx = chunk * 256 + threadIdx.x;
    if (chunk >= nchunk || idx >= total) continue;
    const unsigned u0 = (unsigned)idx;                       // launcher checks total < 2^31
    const unsigned pp = u0 / (unsigned)cpp;
    const int cc = (int)(u0 - pp * (unsigned)cpp);
    const unsigned rowb = pp / (unsigned)nbj;
    const int bj = (int)(pp - rowb * (unsigned)nbj);
    const unsigned vv = rowb / (unsigned)nbi;
    const int bi = (int)(rowb - vv * (unsigned)nbi);
    const int rbase = (BR / 2) * bi - 1, cbase = (BC / 2) * bj - 1;
    float wy[BR + 2][2], wx[BC + 2][2];
#pragma unroll
    for (int k = 0; k < BR + 2; ++k) {
      const int u = BR * bi - 1 + k;
      const float rel = __fsub_rn(__fmul_rn(sy, (float)u), (float)(rbase + (k >> 1)));
      const bool ok = u >= 0 && u < Ho;
      wy[k][1] = ok ? rel : 0.f;
      wy[k][0] = ok ? 1.f - rel : 0.f;
    }
#pragma unroll
    for (int k = 0; k < BC + 2; ++k) {
      const int u = BC * bj - 1 + k;
      const float rel = __fsub_rn(__fmul_rn(sx, (float)u), (float)(cbase + (k >> 1)));
      const bool ok = u >= 0 && u < Wo;
      wx[k][1] = ok ? rel : 0.f;
      wx[k][0] = ok ? 1.f - rel : 0.f;
    }
    int roff[NR], coff[NC];
#pragma unroll
    for (int j = 0; j < NR; ++j) roff[j] = min(max(rbase + j, 0), h - 1) * w;
#pragma unroll
    for (int j = 0; j < NC; ++j) coff[j] = min(max(cbase + j, 0), w - 1);
    const T* zb = z + ((long long)vv * h * w) * ldz + cc * E;

    float acc[BR][BC][E];
    {
      float bv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) bv[e] = bias ? bias[cc * E + e] : 0.f;
#pragma unroll
      for (int a = 0; a < BR; ++a)
#pragma unroll
        for (int b = 0; b < BC; ++b)
#pragma unroll
          for (int e = 0; e < E; ++e) acc[a][b][e] = bv[e];
    }
    // two register sets of z chunks: tap t+1 is requested before tap t is combined.  The scheduling barriers keep that order —
    // left alone, hipcc hoists all 70 loads of a block to the top (280 registers: spills and one wave per SIMD)
    raw_t ra[NR][NC], rb[NR][NC];
#define UP_STEP(KH0, KW0, KH1, KW1, CUR, NXT)                                        \
    UpTap<T, BR, BC, KH1, KW1>::load(zb, Co, ldz, roff, coff, NXT);                 \
    __builtin_amdgcn_sched_barrier(0);                                               \
    UpTap<T, BR, BC, KH0, KW0>::accumulate(CUR, wy, wx, acc);                        \
    __builtin_amdgcn_sched_barrier(0);
    UpTap<T, BR, BC, 0, 0>::load(zb, Co, ldz, roff, coff, ra);
    UP_STEP(0, 0, 0, 1, ra, rb)
    UP_STEP(0, 1, 0, 2, rb, ra)
    UP_STEP(0, 2, 1, 0, ra, rb)
    UP_STEP(1, 0, 1, 1, rb, ra)
    UP_STEP(1, 1, 1, 2, ra, rb)
    UP_STEP(1, 2, 2, 0, rb, ra)
    UP_STEP(2, 0, 2, 1, ra, rb)
    UP_STEP(2, 1, 2, 2, rb, ra)
#undef UP_STEP
    UpTap<T, BR, BC, 2, 2>::accumulate(ra, wy, wx, acc);
    T* ob = out + ((long long)vv * Ho * Wo) * ldo + cc * E;
#pragma unroll
    for (int a = 0; a < BR; ++a) {
      const int oy = BR * bi + a;
#pragma unroll
      for (int b = 0; b < BC; ++b) {
        const int ox = BC * bj + b;      // always inside: the launcher requires 2h % BR == 0 and 2w % BC == 0 (conditional stores made
                                         // hipcc sink each block position's accumulation chain into its store's branch)
        float r[E];
#pragma unroll
        for (int e = 0; e < E; ++e) r[e] = acc[a][b][e] < 0.f ? __builtin_fmaxf(acc[a][b][e], -3.402823466e38f) * nslope : acc[a][b][e];      // NaN stays NaN
        store4(ob + ((long long)oy * Wo + ox) * ldo, r);
      }
    }
  }
}

template <typename T>
int launch_t(const void* z, const float* bias, void* out, int V, int h, int w, int Co, int ldo, float nslope, float sy, float sx,
             hipStream_t s) {
#ifndef UPC_BC
#define UPC_BC 4
#endif
  constexpr int BR = 2, BC = UPC_BC, E = 4;
  const long long total = (long long)V * ((2 * h + BR - 1) / BR) * ((2 * w + BC - 1) / BC) * (Co / E);
  RGBM_REQUIRE(total > 0 && total < (1ll << 31), "upconv combine grid out of range");
  RGBM_REQUIRE((2 * h) % BR == 0 && (2 * w) % BC == 0, "upconv combine: output size must be a multiple of the thread block shape");
  const long long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < (1ll << 20) ? blocks : (1ll << 20));
  hipLaunchKernelGGL((upconv_combine_kernel<T, BR, BC>), dim3(grid), dim3(256), 0, s, (const T*)z, bias, (T*)out, V, h, w, Co, ldo,
                     nslope, sy, sx);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// z [V][h][w][9*Co] (tap-major channel blocks) -> out [V][2h][2w][ldo] (first Co channels) = act(bias + sum over taps)
int launch_upconv_combine(int dtype, const void* z, const float* bias, void* out, int V, int h, int w, int Co, int ldo, int act,
                          float slope, hipStream_t s) {
  RGBM_REQUIRE(Co % 4 == 0 && ldo % dtype_chunk(dtype) == 0 && h >= 2 && w >= 2, "upconv combine geometry");
  RGBM_REQUIRE(act == ACT_NONE || act == ACT_RELU || act == ACT_PRELU, "upconv combine activation");
  const float sy = (float)(h - 1) / (float)(2 * h - 1), sx = (float)(w - 1) / (float)(2 * w - 1);
  const double nout = (double)V * 4.0 * h * w * Co;
  // profiler rows 37 (16-bit storage) / 38 (4-byte slots); algorithmic flops: 36 multiply-adds per output element; bytes: z in + y out
  prof_begin_launch(s, dtype_size(dtype) == 2 ? 37 : 38, 72.0 * nout, ((double)V * h * w * 9.0 * Co + nout) * (double)dtype_size(dtype));
  // one branch-free activation: y = v < 0 ? max(v, -FLT_MAX) * nslope : v (none: 1, ReLU: 0, PReLU: its slope; NaN stays NaN, -inf
  // under ReLU gives -0 instead of NaN, the form conv_igemm_glds.hip uses); per-element branches on `act`
  // made hipcc sink every accumulation chain into the epilogue (all interpolated values live at once)
  const float nslope = act == ACT_NONE ? 1.f : act == ACT_RELU ? 0.f : slope;
  int rc;
  if (dtype == BF16) rc = launch_t<unsigned short>(z, bias, out, V, h, w, Co, ldo, nslope, sy, sx, s);
  else if (dtype == F16) rc = launch_t<f16_t>(z, bias, out, V, h, w, Co, ldo, nslope, sy, sx, s);
  else if (dtype == BF16X3) rc = launch_t<bx3_t>(z, bias, out, V, h, w, Co, ldo, nslope, sy, sx, s);
  else rc = launch_t<float>(z, bias, out, V, h, w, Co, ldo, nslope, sy, sx, s);
  prof_end_launch(s);
  return rc;
}

}  // namespace rgbm
