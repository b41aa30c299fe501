// PSPNet tail in one kernel: up_3 (pspnet.py:100-107: x2 bilinear align_corners -> conv3x3 64 -> 64 + bias -> PReLU) followed by
// `final` (pspnet.py:136: conv1x1 64 -> 32 + bias), from the 64-channel half-resolution tensor to the 32-channel feature map.
//
// up_3 is evaluated in its commuted form (upconv.hip: conv3x3(U x) = sum over taps of shift_t(U(W_t . x))), but the tap-stacked
// tensor z = W_t . x (9 x 64 channels at the low resolution, 7.4 GB per launch at batch 256 in 16-bit storage) never leaves the
// CU: a workgroup owns a 16 x 16 output tile, loads the 10 x 10 low-resolution pixels under it (one pixel of halo) into LDS, and
// per tap runs   z_t = W_t . X   on the matrix pipe (64 x 112 x 64, fp32 result into LDS),   y += shift_t(U z_t)   on the
// vector pipe (the same straight-line pattern-relative interpolation as upconv_combine_kernel, 4 x 4 output pixels x 4 channels
// per thread).  After the nine taps, y + bias -> PReLU is written to LDS in the storage type and `final` is one more small
// matrix product per tile (32 x 256 x 64).  HBM sees the low-resolution input once (plus halo, mostly L2 hits) and the feature
// map once.  Neither the up-sampled tensor, nor z, nor the 64-channel up_3 output exist in memory.
//
// Fixed channel counts (64 -> 64 -> 32: the PSPNet tail); 16-bit storage (bf16 / fp16) and bf16 split pairs.
//
// Round 5, the instantiation bf16 nets run (<bf16, OUTK = 2>: the one that writes the f16 feature map): everything behind the per-tap
// products is f16 - z goes to LDS as f16, the tap combination is packed-f16 FMAs on two channels per instruction (combine_tap_h), PReLU
// is an integer select of {slope, 1} by the sign bit and one packed multiply, y is an f16 MFMA operand, `final` runs on f16 weights.
// Why: fp32 FMAs of one wave do not issue while another wave's MFMAs occupy their SIMD's matrix pipe, packed-f16 ones do
// (tools/micro/mfma_valu_coissue.hip) - the fp32 form's combination (1.1 ms) and per-tap products (0.45 ms) added up without overlap.
// 3.07 -> 2.20 ms per step, and the result is CLOSER to the fp32 reference than the fp32-combining bf16 form (mean error 7.5e-4 against
// 2.2e-3: y and the feature map carry f16's 11 bits instead of bf16's 8; test_upsample_conv3x3_final_fused).
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace rgbm {

namespace {

constexpr int kC = 64;            // channels in / up_3 channels
constexpr int kC2 = 32;           // `final` channels
constexpr int kLow = 10;          // low-resolution pixels per tile side (8 + halo)
constexpr int kZRow = 272;        // bytes per low-resolution pixel of a tap's fp32 z (64 x 4 + 16: bank spread)
constexpr int kNT = 7;            // 16-pixel column tiles of the per-tap product (100 pixels used of 112)

template <typename T> struct Mma16;
template <> struct Mma16<unsigned short> {
  __device__ static __forceinline__ f32x4 run(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma16<f16_t> {
  __device__ static __forceinline__ f32x4 run(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// One tap's contribution to a thread's 4 x 4 output pixels x 4 channels.  zl: the tap's z in LDS at the thread's first
// low-resolution pixel and channel; rows / columns as in upconv.hip (UpTap), block shape 4 x 4.
template <int KH, int KW>
__device__ __forceinline__ void combine_tap(const unsigned char* zl, const float (&wy)[6][2], const float (&wx)[6][2],
                                            f32x4 (&acc)[4][4]) {
  constexpr int JMIN = KH >> 1, JMAX = ((KH + 3) >> 1) + 1;
  constexpr int CMIN = KW >> 1, CMAX = ((KW + 3) >> 1) + 1;
#pragma unroll
  for (int j = JMIN; j <= JMAX; ++j) {
    f32x4 p[4];
#pragma unroll
    for (int c = CMIN; c <= CMAX; ++c) p[c] = *reinterpret_cast<const f32x4*>(zl + (j * kLow + c) * kZRow);
    f32x4 hx[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int c0 = (b + KW) >> 1;
      hx[b] = wx[b + KW][0] * p[c0] + wx[b + KW][1] * p[c0 + 1];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int ra = (a + KH) >> 1;
      if (j != ra && j != ra + 1) continue;
      const float coef = wy[a + KH][j == ra ? 0 : 1];
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += coef * hx[b];
    }
  }
}

// The same in packed f16 (round 5; the bf16 nets' instantiation that writes the f16 feature map): z, the interpolation weights (as {w, w}
// pairs) and the accumulators are f16 pairs, two channels per instruction - half the instructions of the fp32 form, and of the kind that
// issues next to other waves' MFMAs (tools/micro/mfma_valu_coissue.hip), which fp32 FMAs do not: the fp32 form's combination and the
// per-tap products added up without overlap.  acc[a][b] = 4 channels = 2 dwords.
typedef _Float16 uf_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned uf_pk_fma(unsigned a, unsigned b, unsigned c) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_fma(__builtin_bit_cast(uf_h2, a), __builtin_bit_cast(uf_h2, b), __builtin_bit_cast(uf_h2, c)));
}
__device__ __forceinline__ unsigned uf_pk_mul(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(uf_h2, a) * __builtin_bit_cast(uf_h2, b));
}
constexpr int kZRowH = 144;       // bytes per low-resolution pixel of a tap's f16 z (64 x 2 + 16)
template <int KH, int KW>
__device__ __forceinline__ void combine_tap_h(const unsigned char* zl, const unsigned (&wy)[6][2], const unsigned (&wx)[6][2], uint2 (&acc)[4][4]) {
  constexpr int JMIN = KH >> 1, JMAX = ((KH + 3) >> 1) + 1;
  constexpr int CMIN = KW >> 1, CMAX = ((KW + 3) >> 1) + 1;
#pragma unroll
  for (int j = JMIN; j <= JMAX; ++j) {
    uint2 p[4];
#pragma unroll
    for (int c = CMIN; c <= CMAX; ++c) p[c] = *reinterpret_cast<const uint2*>(zl + (j * kLow + c) * kZRowH);
    uint2 hx[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int c0 = (b + KW) >> 1;
      hx[b].x = uf_pk_fma(wx[b + KW][1], p[c0 + 1].x, uf_pk_mul(wx[b + KW][0], p[c0].x));
      hx[b].y = uf_pk_fma(wx[b + KW][1], p[c0 + 1].y, uf_pk_mul(wx[b + KW][0], p[c0].y));
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int ra = (a + KH) >> 1;
      if (j != ra && j != ra + 1) continue;
      const unsigned coef = wy[a + KH][j == ra ? 0 : 1];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        acc[a][b].x = uf_pk_fma(coef, hx[b].x, acc[a][b].x);
        acc[a][b].y = uf_pk_fma(coef, hx[b].y, acc[a][b].y);
      }
    }
  }
}

#ifndef UF_HF_WAVES
#define UF_HF_WAVES 2     // f16 tail: minimum waves per SIMD asked of the register allocator (3 / 4 without UF_HF_XLDS spill: 2.5 -> 2.9 / 7.6 ms)
#endif
#ifndef UF_HF_XLDS
#define UF_HF_XLDS 1      // f16 tail: the tile's B operands are re-read from LDS for every tap instead of living in 56 registers: 191 -> 120
                          // VGPRs, four workgroups per CU instead of two, 2.50 -> 2.20 ms per step (same box, tools/kernel_ms.py)
#endif
template <typename T, int OUTK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OUTK == 2 ? UF_HF_WAVES : 2, 8))) void upconv_final_kernel(const T* __restrict__ x, const T* __restrict__ wz,
                                                           const float* __restrict__ bias, float nslope, const T* __restrict__ wf,
                                                           const float* __restrict__ biasf, void* __restrict__ out, int V, int h,
                                                           int w, float sy, float sx) {
  constexpr bool X3 = std::is_same<T, bx3_t>::value;
  constexpr bool HF = OUTK == 2;                   // f16 tail: z, tap combination, PReLU, Y and `final` in f16 (wf then points at f16 weights)
  constexpr int ZROW = HF ? kZRowH : kZRow;
  if constexpr (HF) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);      // MODE.FP16_OVFL: an overflowing f16 result is +-65504, not inf
  constexpr int EB = (int)sizeof(T);
  constexpr int XROW = kC * EB + 16;               // bytes per pixel row of X (and of Y): 144 / 272
  constexpr int CH = kC * EB / 16;                 // 16-byte chunks per pixel: 8 / 16
  constexpr int NLD = (kLow * kLow * CH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Xs = lds;                         // [112][XROW]  low-resolution tile (rows >= 100 unused)
  unsigned char* Zs = lds + 16 * kNT * XROW;       // [112][kZRow] one tap's z, fp32
  unsigned char* Ys = lds;                         // [256][XROW]  up_3 output of the tile (after the taps; aliases X and Z)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int Ho = 2 * h, Wo = 2 * w, nty = Ho / 16, ntx = Wo / 16;
  // every XCD (own L2) takes one contiguous eighth of the tiles: neighbours share their halo pixels
  const unsigned ntile = (unsigned)(V * nty * ntx), per_xcd = (ntile + 7) / 8;
  const unsigned tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= ntile) return;
  const unsigned v = tile / (unsigned)(nty * ntx);
  const unsigned rem = tile - v * (unsigned)(nty * ntx);
  const int ty = (int)(rem / (unsigned)ntx), tx = (int)(rem - (unsigned)ty * (unsigned)ntx);

  // ---- low-resolution tile -> LDS (clamped at the image border: those pixels get weight 0 or O(1e-7)) ----
  {
    uint4 val[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      const int px = min(i / CH, kLow * kLow - 1), c = i & (CH - 1);
      const int pr = px / kLow, pc = px - pr * kLow;
      const int gr = min(max(8 * ty - 1 + pr, 0), h - 1), gc = min(max(8 * tx - 1 + pc, 0), w - 1);
      val[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(x) +
                                               (((long long)v * h + gr) * w + gc) * (kC * EB) + c * 16);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * 256;
      if (i < kLow * kLow * CH) *reinterpret_cast<uint4*>(Xs + (i / CH) * XROW + (i & (CH - 1)) * 16) = val[k];
    }
  }

  // ---- the thread's 4 x 4 output block, 4 channels: interpolation weights relative to the fixed row / column pattern ----
  const int g = tid & 15, blk = tid >> 4, bi = blk >> 2, bj = blk & 3;
  float wy[6][2], wx[6][2];
  {
    const int BI = 4 * ty + bi, BJ = 4 * tx + bj;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int u = 4 * BI - 1 + k;
      const float rel = __fsub_rn(__fmul_rn(sy, (float)u), (float)(2 * BI - 1 + (k >> 1)));
      const bool ok = u >= 0 && u < Ho;
      wy[k][1] = ok ? rel : 0.f;
      wy[k][0] = ok ? 1.f - rel : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int u = 4 * BJ - 1 + k;
      const float rel = __fsub_rn(__fmul_rn(sx, (float)u), (float)(2 * BJ - 1 + (k >> 1)));
      const bool ok = u >= 0 && u < Wo;
      wx[k][1] = ok ? rel : 0.f;
      wx[k][0] = ok ? 1.f - rel : 0.f;
    }
  }
  f32x4 acc[HF ? 1 : 4][HF ? 1 : 4];
  uint2 acch[HF ? 4 : 1][HF ? 4 : 1];
  unsigned wyh[6][2], wxh[6][2];
  auto pair16 = [](float a, float b) { const uf_h2 h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); };
  {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * g);
    if constexpr (HF) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acch[a][b] = make_uint2(pair16(bv[0], bv[1]), pair16(bv[2], bv[3]));
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) { wyh[k][e] = pair16(wy[k][e], wy[k][e]); wxh[k][e] = pair16(wx[k][e], wx[k][e]); }
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = bv;
    }
  }
  const unsigned char* zl = Zs + ((2 * bi) * kLow + 2 * bj) * ZROW + g * (HF ? 8 : 16);
  unsigned char* zw = Zs + lr * ZROW + (16 * wv + 4 * lg) * (HF ? 2 : 4);       // this lane's slot of column tile 0 (wave wv: channels 16 wv ..)

  __syncthreads();
  // 16-bit storage: the tile's B operands stay in registers for all nine taps (7 column tiles x 2 K steps)
  constexpr bool XREG = !X3 && !(HF && UF_HF_XLDS);
  uint4 xf[XREG ? kNT : 1][2];
  if constexpr (XREG) {
#pragma unroll
    for (int n = 0; n < kNT; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) xf[n][ks] = *reinterpret_cast<const uint4*>(Xs + (n * 16 + lr) * XROW + ks * 64 + lg * 16);
  }

  // A operand of a tap: rows 16 wv .. 16 wv + 15 of the tap's 64 x 64 weight block ([9 * 64][64] row-major)
  uint4 af[X3 ? 4 : 2];
  auto load_a = [&](int t) {
    const T* row = wz + ((long long)(t * kC + 16 * wv + lr)) * kC + lg * 8;
    if constexpr (X3) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        af[2 * ks] = *reinterpret_cast<const uint4*>(row + ks * 32);
        af[2 * ks + 1] = *reinterpret_cast<const uint4*>(row + ks * 32 + 4);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[ks] = *reinterpret_cast<const uint4*>(row + ks * 32);
    }
  };
  load_a(0);

#define UF_TAP(KH, KW)                                                                                        \
  {                                                                                                           \
    f32x4 zacc[kNT];                                                                                          \
    _Pragma("unroll") for (int n = 0; n < kNT; ++n) zacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};                     \
    if constexpr (X3) {                                                                                       \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                      \
        uint4 ah, al, bh[kNT], bl[kNT];                                                                       \
        bx3_pair(af[2 * ks], af[2 * ks + 1], ah, al);                                                         \
        _Pragma("unroll") for (int n = 0; n < kNT; ++n) {                                                     \
          const unsigned char* p = Xs + (n * 16 + lr) * XROW + ks * 128 + lg * 32;                            \
          bx3_pair(*reinterpret_cast<const uint4*>(p), *reinterpret_cast<const uint4*>(p + 16), bh[n], bl[n]); \
        }                                                                                                     \
        _Pragma("unroll") for (int n = 0; n < kNT; ++n) zacc[n] = Mma16<unsigned short>::run(al, bh[n], zacc[n]); \
        _Pragma("unroll") for (int n = 0; n < kNT; ++n) zacc[n] = Mma16<unsigned short>::run(ah, bl[n], zacc[n]); \
        _Pragma("unroll") for (int n = 0; n < kNT; ++n) zacc[n] = Mma16<unsigned short>::run(ah, bh[n], zacc[n]); \
      }                                                                                                       \
    } else if constexpr (!XREG) {                                                                             \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
        _Pragma("unroll") for (int n = 0; n < kNT; ++n)                                                       \
          zacc[n] = Mma16<typename std::conditional<X3, unsigned short, T>::type>::run(                      \
              af[ks], *reinterpret_cast<const uint4*>(Xs + (n * 16 + lr) * XROW + ks * 64 + lg * 16), zacc[n]); \
    } else {                                                                                                  \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
        _Pragma("unroll") for (int n = 0; n < kNT; ++n)                                                       \
          zacc[n] = Mma16<typename std::conditional<X3, unsigned short, T>::type>::run(af[ks], xf[n][ks], zacc[n]); \
    }                                                                                                         \
    if (KH * 3 + KW < 8) load_a(KH * 3 + KW + 1);                                                             \
    __syncthreads(); /* the previous tap's z has been read */                                                 \
    if constexpr (HF) {                                                                                       \
      _Pragma("unroll") for (int n = 0; n < kNT; ++n)                                                         \
        *reinterpret_cast<uint2*>(zw + n * 16 * ZROW) = make_uint2(pair16(zacc[n][0], zacc[n][1]), pair16(zacc[n][2], zacc[n][3])); \
    } else {                                                                                                  \
      _Pragma("unroll") for (int n = 0; n < kNT; ++n) *reinterpret_cast<f32x4*>(zw + n * 16 * kZRow) = zacc[n]; \
    }                                                                                                         \
    __syncthreads();                                                                                          \
    if constexpr (HF) combine_tap_h<KH, KW>(zl, wyh, wxh, acch);                                              \
    else combine_tap<KH, KW>(zl, wy, wx, acc);                                                                \
  }
  UF_TAP(0, 0) UF_TAP(0, 1) UF_TAP(0, 2)
  UF_TAP(1, 0) UF_TAP(1, 1) UF_TAP(1, 2)
  UF_TAP(2, 0) UF_TAP(2, 1) UF_TAP(2, 2)
#undef UF_TAP

  // ---- PReLU -> Y (storage type) in LDS ----
  __syncthreads();      // X and the last z are dead
  if constexpr (HF) {
    // PReLU on the packed pairs: factor = the sign bit selects {slope, 1} per half (integer instructions), y = x * factor - NaN stays NaN
    typedef short uf_s2 __attribute__((ext_vector_type(2)));
    const unsigned one2 = 0x3c003c00u, sl2 = pair16(nslope, nslope);
    auto prelu2 = [&](unsigned x) {
      const unsigned m = __builtin_bit_cast(unsigned, __builtin_bit_cast(uf_s2, x) >> 15);      // all ones in a negative half
      return uf_pk_mul(x, (m & sl2) | (~m & one2));
    };
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int p = (4 * bi + a) * 16 + 4 * bj + b;
        *reinterpret_cast<uint2*>(Ys + p * XROW + 8 * g) = make_uint2(prelu2(acch[a][b].x), prelu2(acch[a][b].y));
      }
  } else {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = acc[a][b][e] < 0.f ? __builtin_fmaxf(acc[a][b][e], -3.402823466e38f) * nslope : acc[a][b][e];
      const int p = (4 * bi + a) * 16 + 4 * bj + b;
      store4(reinterpret_cast<T*>(Ys + p * XROW) + 4 * g, r);
    }
  }
  __syncthreads();

  // ---- final 1x1: feat[32][256] = Wf[32][64] . Y^T; wave wv takes output rows 4 wv .. 4 wv + 3 of the tile ----
  f32x4 facc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) facc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if constexpr (X3) {
      uint4 ah[2], al[2], bh[4], bl[4];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const T* row = wf + (m * 16 + lr) * kC + ks * 32 + lg * 8;
        bx3_pair(*reinterpret_cast<const uint4*>(row), *reinterpret_cast<const uint4*>(row + 4), ah[m], al[m]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned char* p = Ys + ((4 * wv + q) * 16 + lr) * XROW + ks * 128 + lg * 32;
        bx3_pair(*reinterpret_cast<const uint4*>(p), *reinterpret_cast<const uint4*>(p + 16), bh[q], bl[q]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) facc[m][q] = Mma16<unsigned short>::run(al[m], bh[q], facc[m][q]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) facc[m][q] = Mma16<unsigned short>::run(ah[m], bl[q], facc[m][q]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) facc[m][q] = Mma16<unsigned short>::run(ah[m], bh[q], facc[m][q]);
    } else {
      uint4 a2[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) a2[m] = *reinterpret_cast<const uint4*>(wf + (m * 16 + lr) * kC + ks * 32 + lg * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 b2 = *reinterpret_cast<const uint4*>(Ys + ((4 * wv + q) * 16 + lr) * XROW + ks * 64 + lg * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m)
          facc[m][q] = Mma16<typename std::conditional<X3, unsigned short, typename std::conditional<HF, f16_t, T>::type>::type>::run(a2[m], b2, facc[m][q]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(biasf + m * 16 + lg * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = 16 * ty + 4 * wv + q, ox = 16 * tx + lr;
      const long long o = (((long long)v * Ho + oy) * Wo + ox) * kC2 + m * 16 + lg * 4;
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = facc[m][q][e] + bv[e];
      if constexpr (OUTK == 1) store4(reinterpret_cast<float*>(out) + o, r);
      else if constexpr (OUTK == 2) store4(reinterpret_cast<f16_t*>(out) + o, r);      // saturating
      else store4(reinterpret_cast<T*>(out) + o, r);
    }
  }
}

template <typename T, int OUTK>
int launch_t(const void* x, const void* wz, const float* bias, float nslope, const void* wf, const float* biasf, void* out, int V,
             int h, int w, hipStream_t s) {
  constexpr int XROW = kC * (int)sizeof(T) + 16;
  constexpr int lds_taps = 16 * kNT * XROW + 16 * kNT * (OUTK == 2 ? kZRowH : kZRow), lds_y = 256 * XROW;
  constexpr int lds = lds_taps > lds_y ? lds_taps : lds_y;
  auto kern = upconv_final_kernel<T, OUTK>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
  const long long ntile = (long long)V * (2 * h / 16) * (2 * w / 16);
  RGBM_REQUIRE(ntile > 0 && ntile < (1ll << 30), "upconv + final grid out of range");
  const unsigned grid = (unsigned)(((ntile + 7) / 8) * 8);
  const float sy = (float)(h - 1) / (float)(2 * h - 1), sx = (float)(w - 1) / (float)(2 * w - 1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, (const T*)x, (const T*)wz, bias, nslope, (const T*)wf, biasf, out, V, h, w,
                     sy, sx);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// x [V][h][w][64] -> out [V][2h][2w][32] = final(PReLU(up_3(x)));  wz [9 * 64][64] (tap-major rows, storage type), wf [32][64].
// out_f32 (the output kind): 0 = storage type; 1 = the feature map as plain fp32 (bf16x3 nets: the split-pair sweep kernel reads it so);
// 2 = f16 (bf16 nets: the plane sweep's packed-f16 blend and the point heads read it so - AdaPose::feat_f16()).
int launch_upconv_final(int dtype, const void* x, const void* wz, const float* bias, float slope, const void* wf, const float* biasf,
                        void* out, int out_f32, int V, int h, int w, hipStream_t s, const void* wf_f16) {
  RGBM_REQUIRE(dtype == BF16 || dtype == F16 || dtype == BF16X3, "upconv + final: 16-bit or split-pair storage");
  RGBM_REQUIRE(h >= 8 && w >= 8 && h % 8 == 0 && w % 8 == 0, "upconv + final: low-resolution size must be a multiple of 8");
  RGBM_REQUIRE(out_f32 >= 0 && out_f32 <= 2 && (out_f32 != 1 || dtype == BF16X3) && (out_f32 != 2 || dtype == BF16),
               "upconv + final: plain fp32 output is the split-pair path's, f16 output the bf16 path's");
  const double npx = (double)V * 4.0 * h * w;
  // profiler row 39; algorithmic flops: the commuted product (9 x 64 x 64 per low-resolution pixel), 36 multiply-adds per up_3
  // output element, the 1x1; bytes: x in + feature map out
  prof_begin_launch(s, 39, 2.0 * (npx / 4.0) * 9.0 * kC * kC + 72.0 * npx * kC + 2.0 * npx * kC * kC2,
                    (npx / 4.0) * kC * dtype_size(dtype) + npx * kC2 * (out_f32 == 1 ? 4.0 : (double)dtype_size(dtype)));
  int rc;
  RGBM_REQUIRE(out_f32 != 2 || wf_f16 != nullptr, "upconv + final: the f16 tail needs `final`'s weights in f16");
  if (dtype == BF16 && out_f32 == 2) rc = launch_t<unsigned short, 2>(x, wz, bias, slope, wf_f16, biasf, out, V, h, w, s);
  else if (dtype == BF16) rc = launch_t<unsigned short, 0>(x, wz, bias, slope, wf, biasf, out, V, h, w, s);
  else if (dtype == F16) rc = launch_t<f16_t, 0>(x, wz, bias, slope, wf, biasf, out, V, h, w, s);
  else if (out_f32) rc = launch_t<bx3_t, 1>(x, wz, bias, slope, wf, biasf, out, V, h, w, s);
  else rc = launch_t<bx3_t, 0>(x, wz, bias, slope, wf, biasf, out, V, h, w, s);
  prof_end_launch(s);
  return rc;
}

}  // namespace rgbm
