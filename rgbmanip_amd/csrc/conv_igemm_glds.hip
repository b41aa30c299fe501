// Implicit-GEMM convolution, LDS-DMA variant (gfx950): same GEMM view, tiles, fragment maps and epilogue as
// conv_igemm.hip, but both operand tiles travel global -> LDS with `global_load_lds_dwordx4` (no VGPR staging, no
// ds_write pass), so the loads of K tile t+1 are in flight while the MFMAs of tile t run and only one barrier per K tile
// is needed.  The LDS image of a tile is lane-linear (wave-uniform base + lane*16 bytes), therefore the bank-conflict
// swizzle is applied on the SOURCE side: the lane that fills LDS slot (row, j) fetches global chunk j ^ ((row>>1)&7),
// and readers XOR the same value (an involution).  Out-of-image / padded-K lanes fetch from a 16-byte zero page.
#include "common.h"
#include "prof.h"

#ifndef IG_ABL
#define IG_ABL 0      // ablation builds only (tools/abl_build.sh): 1 = pixel gathers read the zero page, 2 = no MFMA, 4 = weights too, 8 = no DMA
#endif

namespace rgbm {

extern int g_debug_flags;
__device__ uint4 g_zero_page[4];     // zero-initialised device memory: the source of every padded chunk

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// LDS-DMA issued from inline asm ON PURPOSE: for the builtin, hipcc (ROCm 7.2) conservatively places `s_waitcnt vmcnt(0)`
// in front of the first ds_read that follows, which drains the loads of the NEXT K tile before the current one is
// multiplied (seen in the ISA: issue -> vmcnt(0) -> ds_read -> mfma, i.e. no load/compute overlap inside a workgroup).
// An asm DMA is invisible to that bookkeeping; completion is enforced by our own counted `s_waitcnt vmcnt(N)` + barrier.
// lds_off: wave-uniform LDS byte address of this wave's 1 KiB slot (hardware adds lane*16).  M0 is left modified: nothing
// else in these kernels reads it (gfx9+ DS instructions do not), and saving/restoring it cost 2 SALU per piece.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_off) {
#if IG_ABL & 8
  asm volatile("" :: "v"(gsrc), "s"(lds_off) : "memory");
  return;
#endif
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_off) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

template <typename T> struct MmaG;
template <> struct MmaG<unsigned short> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct MmaG<float> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

__device__ __forceinline__ float apply_act_g(float v, int act, float slope) {
  if (act == ACT_RELU) return v < 0.f ? 0.f : v;          // NaN propagates like torch
  if (act == ACT_PRELU) return v < 0.f ? v * slope : v;
  if (act == ACT_TANH) return tanhf(v);
  return v;
}

// UNI: every K tile lies inside ONE filter tap (Cin is a multiple of BK, or the conv is 1x1), so the tap walk is
// wave-uniform scalar work and a gathered row costs ~6 VALU per K tile: per-row source pointer of tap (0,0,0) and a
// (kd | kh<<8 | kw<<16) validity bit mask are built once in the prologue; per tile the lane adds a scalar byte
// offset and tests the mask.  (The general path recomputes tap -> (kd,kh,kw) -> 64-bit address per lane per tile,
// ~35 VALU with quarter-rate integer multiplies per row: measured, it cost more issue time than the MFMAs.)
template <typename T, int BCH, int BPIX, bool UNI>
__global__ __launch_bounds__(256, 2) void conv_igemm_glds_kernel(const ConvDesc d) {
  constexpr int E = 16 / sizeof(T);
  constexpr int BK = 8 * E;
  constexpr int XR = BPIX / 32;
  constexpr int WL = BCH >= 32 ? BCH / 32 : 1;
  constexpr int WCH = BCH < 64 ? BCH : 64;
  constexpr int FM = WCH / 16;
  constexpr int FN = 4;
  constexpr int STAGE = (BCH + BPIX) * 8;   // uint4 slots per stage
  __shared__ uint4 lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // SGPR: keeps the per-wave bookkeeping scalar

  const int nblk = gridDim.x;
  const int bq = nblk >> 3, br = nblk & 7;
  const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  const int lid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
  const int pix_tile = lid / d.n_ch_tiles;
  const int ch_tile = lid - pix_tile * d.n_ch_tiles;

  const T* __restrict__ in = reinterpret_cast<const T*>(d.in);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(d.wgt);

  // LDS slot (row, j) is filled by thread (r0 = row & 31 [+32*i], j); it fetches source chunk js = j ^ swizzle(row).
  const int j = tid & 7;
  const int r0 = tid >> 3;
  const int js = j ^ ((r0 >> 1) & 7);       // ((r0 + 32*i) >> 1) & 7 == (r0 >> 1) & 7
  int xn[XR], xd0[XR], xh0[XR], xw0[XR];
  const char* rowp[XR];
  unsigned rmask[XR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const long long m = (long long)pix_tile * BPIX + r0 + 32 * i;
    if (m < d.M) {
      unsigned t = (unsigned)m;
      const unsigned qw = t % (unsigned)d.Wq; t /= (unsigned)d.Wq;
      const unsigned qh = t % (unsigned)d.Hq; t /= (unsigned)d.Hq;
      const unsigned qd = t % (unsigned)d.Dq; t /= (unsigned)d.Dq;
      xn[i] = (int)t * d.Di;
      xd0[i] = (int)qd * d.sd - d.pd;
      xh0[i] = (int)qh * d.sh - d.ph;
      xw0[i] = (int)qw * d.sw - d.pw;
    } else {
      xn[i] = 0; xd0[i] = -(1 << 20); xh0[i] = 0; xw0[i] = 0;
    }
    if (UNI) {
      unsigned mk = 0;
      for (int k = 0; k < d.KD; ++k) mk |= (unsigned)((unsigned)(xd0[i] + k * d.dild) < (unsigned)d.Di) << k;
      for (int k = 0; k < d.KH; ++k) mk |= (unsigned)((unsigned)(xh0[i] + k * d.dilh) < (unsigned)d.Hi) << (8 + k);
      for (int k = 0; k < d.KW; ++k) mk |= (unsigned)((unsigned)(xw0[i] + k * d.dilw) < (unsigned)d.Wi) << (16 + k);
      rmask[i] = mk;
      const long long pix0 = ((long long)(xn[i] + xd0[i]) * d.Hi + xh0[i]) * d.Wi + xw0[i];
      rowp[i] = reinterpret_cast<const char*>(in) + (pix0 * d.Cin + js * E) * (long long)sizeof(T);
    }
  }
  const float rcp_khw = 1.0f / (float)(d.KH * d.KW);
  const float rcp_kw = 1.0f / (float)d.KW;
  const int khw = d.KH * d.KW;
  const bool wload = (BCH >= 32) || (wave < BCH / 8);      // wave-uniform
  const T* zero = reinterpret_cast<const T*>(g_zero_page);
  const char* wrow[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i)
    wrow[i] = reinterpret_cast<const char*>(wgt + (long long)(ch_tile * BCH + r0 + 32 * i) * d.Kpad + js * E);
  int tkd = 0, tkh = 0, tkw = 0, tc = 0;     // UNI: wave-uniform tap walker (issue() is called with kt = 0,1,2,...)

  auto issue = [&](int kt, int stage) {
    uint4* W = lds + stage * STAGE;
    uint4* X = W + BCH * 8;
    if (UNI) {
      const unsigned sel = (1u << tkd) | (256u << tkh) | (65536u << tkw);
      const long long soff =
          ((long long)((tkd * d.dild * d.Hi + tkh * d.dilh) * d.Wi + tkw * d.dilw) * d.Cin + tc) * (long long)sizeof(T);
      const bool cok = tkd < d.KD && tc + js * E < d.Cin;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const bool ok = cok && (rmask[i] & sel) == sel;
        const char* src = (ok && !(IG_ABL & 1)) ? rowp[i] + soff : reinterpret_cast<const char*>(zero);
        glds16(src, __builtin_amdgcn_readfirstlane(lds_addr(X + (i * 32 + wave * 8) * 8)));
      }
      tc += BK;
      if (d.lcin >= 0 && tc >= d.Cin) {
        tc = 0;
        if (++tkw == d.KW) { tkw = 0; if (++tkh == d.KH) { tkh = 0; ++tkd; } }
      }
    } else {
      const int k = kt * BK + js * E;
      int tap, c;
      bool tapok;
      if (d.lcin >= 0) { tap = k >> d.lcin; c = k & (d.Cin - 1); tapok = tap < d.ntaps; }
      else { tap = 0; c = k; tapok = k < d.Cin; }
      const int kd = (int)(((float)tap + 0.5f) * rcp_khw);
      const int rem = tap - kd * khw;
      const int kh = (int)(((float)rem + 0.5f) * rcp_kw);
      const int kw = rem - kh * d.KW;
      const int od = kd * d.dild, oh = kh * d.dilh, ow = kw * d.dilw;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const int dd = xd0[i] + od, hh = xh0[i] + oh, ww = xw0[i] + ow;
        const bool ok = tapok && (unsigned)dd < (unsigned)d.Di && (unsigned)hh < (unsigned)d.Hi && (unsigned)ww < (unsigned)d.Wi;
        const long long pix = ((long long)(xn[i] + dd) * d.Hi + hh) * d.Wi + ww;
        const T* src = ok ? in + pix * d.Cin + c : zero;
        // wave-uniform LDS base of this wave's 64 consecutive slots; the hardware adds lane*16 bytes
        glds16(src, __builtin_amdgcn_readfirstlane(lds_addr(X + (i * 32 + wave * 8) * 8)));
      }
    }
    const long long wk = (long long)kt * BK * (long long)sizeof(T);
    if (wload) {
#pragma unroll
      for (int i = 0; i < WL; ++i)
        glds16((IG_ABL & 4) ? reinterpret_cast<const char*>(zero) : wrow[i] + wk, __builtin_amdgcn_readfirstlane(lds_addr(W + (i * 32 + wave * 8) * 8)));
    }
  };

  const int wch = (BCH == 128) ? (wave >> 1) * 64 : 0;
  const int wpix = (BCH == 128) ? (wave & 1) * 64 : wave * 64;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lr = lane & 15, lg = lane >> 4;

  issue(0, 0);
  int cur = 0;
  for (int kt = 0; kt < d.KT; ++kt) {
    // tile kt has landed for every wave (vmcnt(0) + barrier); every wave has also finished reading the other stage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < d.KT) issue(kt + 1, cur ^ 1);
    const uint4* W = lds + cur * STAGE;
    const uint4* X = W + BCH * 8;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cidx = s * 4 + lg;
      uint4 af[FM], bf[FN];
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        const int row = wch + a * 16 + lr;
        af[a] = W[row * 8 + (cidx ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int b = 0; b < FN; ++b) {
        const int row = wpix + b * 16 + lr;
        bf[b] = X[row * 8 + (cidx ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) MmaG<T>::run(af[a], bf[b], acc[a][b]);
    }
    cur ^= 1;
  }

  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(d.res);
#pragma unroll
  for (int b = 0; b < FN; ++b) {
    const long long m = (long long)pix_tile * BPIX + wpix + b * 16 + lr;
    if (m >= d.M) continue;
    unsigned t = (unsigned)m;
    const unsigned qw = t % (unsigned)d.Wq; t /= (unsigned)d.Wq;
    const unsigned qh = t % (unsigned)d.Hq; t /= (unsigned)d.Hq;
    const unsigned qd = t % (unsigned)d.Dq; t /= (unsigned)d.Dq;
    const int n = (int)t;
    const long long opix = (((long long)n * d.Do + (qd * d.osd + d.opd)) * d.Ho + (qh * d.osh + d.oph)) * d.Wo +
                           (qw * d.osw + d.opw);
#pragma unroll
    for (int a = 0; a < FM; ++a) {
      const int ch = ch_tile * BCH + wch + a * 16 + lg * 4;
      if (ch >= d.Cout) continue;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (d.bias) {
        const float* bp = d.bias + (long long)n * d.bias_stride + ch;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bp[e];
      }
      const long long o = opix * d.ldo + ch;
      if (d.res_mode == RES_PRE_ACT) {
        float rv[4];
        load4(res + o, rv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act_g(v[e], d.act, d.slope);
      if (d.res_mode == RES_POST_ACT) {
        float rv[4];
        load4(res + o, rv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
      }
      store4(out + o, v);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// 256 x 128 tile, 8 waves, 3-stage LDS ring with counted vmcnt (layers with >= 128 output channels).
// Two K tiles are in flight while a third is multiplied: tile kt+2 is issued right after the single barrier of
// iteration kt, the wait in front of that barrier is `vmcnt(6)` (the 6 LDS-DMA instructions of tile kt+1 may stay
// outstanding), never 0 inside the loop.  Raw s_barrier + inline waits: __syncthreads() would drain the DMA queue.
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool UNI>
__global__ __launch_bounds__(512, 1) void conv_igemm_v3_kernel(const ConvDesc d) {
  constexpr int BCH = 128, BPIX = 256;
  constexpr int E = 16 / sizeof(T);
  constexpr int BK = 8 * E;
  constexpr int XR = BPIX / 64;              // 4 gathered rows per thread
  constexpr int WL = BCH / 64;               // 2 weight rows per thread
  constexpr int FM = 4, FN = 4;
  constexpr int STAGE = (BCH + BPIX) * 8;    // uint4 slots per stage (48 KB)
  extern __shared__ __attribute__((aligned(16))) uint4 lds3[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7, SGPR

  const int nblk = gridDim.x;
  const int bq = nblk >> 3, br = nblk & 7;
  const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  const int lid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
  const int pix_tile = lid / d.n_ch_tiles;
  const int ch_tile = lid - pix_tile * d.n_ch_tiles;

  const T* __restrict__ in = reinterpret_cast<const T*>(d.in);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(d.wgt);
  const int j = tid & 7;
  const int r0 = tid >> 3;                   // 0..63
  const int js = j ^ ((r0 >> 1) & 7);
  int xn[XR], xd0[XR], xh0[XR], xw0[XR];
  const char* rowp[XR];
  unsigned rmask[XR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const long long m = (long long)pix_tile * BPIX + r0 + 64 * i;
    if (m < d.M) {
      unsigned t = (unsigned)m;
      const unsigned qw = t % (unsigned)d.Wq; t /= (unsigned)d.Wq;
      const unsigned qh = t % (unsigned)d.Hq; t /= (unsigned)d.Hq;
      const unsigned qd = t % (unsigned)d.Dq; t /= (unsigned)d.Dq;
      xn[i] = (int)t * d.Di;
      xd0[i] = (int)qd * d.sd - d.pd;
      xh0[i] = (int)qh * d.sh - d.ph;
      xw0[i] = (int)qw * d.sw - d.pw;
    } else {
      xn[i] = 0; xd0[i] = -(1 << 20); xh0[i] = 0; xw0[i] = 0;
    }
    if (UNI) {
      unsigned mk = 0;
      for (int k = 0; k < d.KD; ++k) mk |= (unsigned)((unsigned)(xd0[i] + k * d.dild) < (unsigned)d.Di) << k;
      for (int k = 0; k < d.KH; ++k) mk |= (unsigned)((unsigned)(xh0[i] + k * d.dilh) < (unsigned)d.Hi) << (8 + k);
      for (int k = 0; k < d.KW; ++k) mk |= (unsigned)((unsigned)(xw0[i] + k * d.dilw) < (unsigned)d.Wi) << (16 + k);
      rmask[i] = mk;
      const long long pix0 = ((long long)(xn[i] + xd0[i]) * d.Hi + xh0[i]) * d.Wi + xw0[i];
      rowp[i] = reinterpret_cast<const char*>(in) + (pix0 * d.Cin + js * E) * (long long)sizeof(T);
    }
  }
  const float rcp_khw = 1.0f / (float)(d.KH * d.KW);
  const float rcp_kw = 1.0f / (float)d.KW;
  const int khw = d.KH * d.KW;
  const T* zero = reinterpret_cast<const T*>(g_zero_page);
  const char* wrow[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i)
    wrow[i] = reinterpret_cast<const char*>(wgt + (long long)(ch_tile * BCH + r0 + 64 * i) * d.Kpad + js * E);
  int tkd = 0, tkh = 0, tkw = 0, tc = 0;     // UNI: wave-uniform tap walker (issue() is called with kt = 0,1,2,...)

  auto issue = [&](int kt, int stage) {
    uint4* W = lds3 + stage * STAGE;
    uint4* X = W + BCH * 8;
    if (UNI) {
      const unsigned sel = (1u << tkd) | (256u << tkh) | (65536u << tkw);
      const long long soff =
          ((long long)((tkd * d.dild * d.Hi + tkh * d.dilh) * d.Wi + tkw * d.dilw) * d.Cin + tc) * (long long)sizeof(T);
      const bool cok = tkd < d.KD && tc + js * E < d.Cin;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const bool ok = cok && (rmask[i] & sel) == sel;
        const char* src = (ok && !(IG_ABL & 1)) ? rowp[i] + soff : reinterpret_cast<const char*>(zero);
        glds16(src, __builtin_amdgcn_readfirstlane(lds_addr(X + (i * 64 + wave * 8) * 8)));
      }
      tc += BK;
      if (d.lcin >= 0 && tc >= d.Cin) {
        tc = 0;
        if (++tkw == d.KW) { tkw = 0; if (++tkh == d.KH) { tkh = 0; ++tkd; } }
      }
    } else {
      const int k = kt * BK + js * E;
      int tap, c;
      bool tapok;
      if (d.lcin >= 0) { tap = k >> d.lcin; c = k & (d.Cin - 1); tapok = tap < d.ntaps; }
      else { tap = 0; c = k; tapok = k < d.Cin; }
      const int kd = (int)(((float)tap + 0.5f) * rcp_khw);
      const int rem = tap - kd * khw;
      const int kh = (int)(((float)rem + 0.5f) * rcp_kw);
      const int kw = rem - kh * d.KW;
      const int od = kd * d.dild, oh = kh * d.dilh, ow = kw * d.dilw;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const int dd = xd0[i] + od, hh = xh0[i] + oh, ww = xw0[i] + ow;
        const bool ok = tapok && (unsigned)dd < (unsigned)d.Di && (unsigned)hh < (unsigned)d.Hi && (unsigned)ww < (unsigned)d.Wi;
        const long long pix = ((long long)(xn[i] + dd) * d.Hi + hh) * d.Wi + ww;
        const T* src = ok ? in + pix * d.Cin + c : zero;
        // wave-uniform LDS base of this wave's 64 consecutive slots; the hardware adds lane*16 bytes
        glds16(src, __builtin_amdgcn_readfirstlane(lds_addr(X + (i * 64 + wave * 8) * 8)));
      }
    }
    const long long wk = (long long)kt * BK * (long long)sizeof(T);
    if (true) {
#pragma unroll
      for (int i = 0; i < WL; ++i)
        glds16((IG_ABL & 4) ? reinterpret_cast<const char*>(zero) : wrow[i] + wk, __builtin_amdgcn_readfirstlane(lds_addr(W + (i * 64 + wave * 8) * 8)));
    }
  };

  const int wch = (wave >> 2) * 64;
  const int wpix = (wave & 3) * 64;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lr = lane & 15, lg = lane >> 4;

  auto compute = [&](int st) {
    const uint4* W = lds3 + st * STAGE;
    const uint4* X = W + BCH * 8;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cidx = s * 4 + lg;
      uint4 af[FM], bf[FN];
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        const int row = wch + a * 16 + lr;
        af[a] = W[row * 8 + (cidx ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int b = 0; b < FN; ++b) {
        const int row = wpix + b * 16 + lr;
        bf[b] = X[row * 8 + (cidx ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) {
          if (!(IG_ABL & 2)) MmaG<T>::run(af[a], bf[b], acc[a][b]);
          else acc[a][b][0] += __uint_as_float(af[a].x ^ bf[b].x);
        }
    }
  };

  issue(0, 0);
  if (d.KT > 1) issue(1, 1);
  int st = 0;                                  // stage of tile kt
  // The two waves of a SIMD run the halves of an iteration in opposite order: waves 0-3 request tile kt+2 (address
  // VALU/SALU + DMA issue) and then multiply tile kt, waves 4-7 multiply first and request afterwards, so one wave's
  // request phase runs under the other's MFMAs instead of both idling the matrix pipe at the same time.  Legal in either
  // order: the stage being refilled was last read in iteration kt-1, i.e. before this iteration's barrier.
  const bool issue_first = wave < 4;            // wave is an SGPR: a scalar branch
  for (int kt = 0; kt < d.KT; ++kt) {
    if (kt + 1 < d.KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // tile kt landed; tile kt+1 may still fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // every wave's share of tile kt is in LDS; stage of tile kt-1 is free
    const int nst = st == 0 ? 2 : st - 1;
    const bool more = kt + 2 < d.KT;
    if (issue_first && more) issue(kt + 2, nst);       // one copy of compute(): two copies made hipcc shuffle the
    compute(st);                                       // accumulators between differently allocated paths
    if (!issue_first && more) issue(kt + 2, nst);
    st = st == 2 ? 0 : st + 1;
  }

  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(d.res);
#pragma unroll
  for (int b = 0; b < FN; ++b) {
    const long long m = (long long)pix_tile * BPIX + wpix + b * 16 + lr;
    if (m >= d.M) continue;
    unsigned t = (unsigned)m;
    const unsigned qw = t % (unsigned)d.Wq; t /= (unsigned)d.Wq;
    const unsigned qh = t % (unsigned)d.Hq; t /= (unsigned)d.Hq;
    const unsigned qd = t % (unsigned)d.Dq; t /= (unsigned)d.Dq;
    const int n = (int)t;
    const long long opix = (((long long)n * d.Do + (qd * d.osd + d.opd)) * d.Ho + (qh * d.osh + d.oph)) * d.Wo +
                           (qw * d.osw + d.opw);
#pragma unroll
    for (int a = 0; a < FM; ++a) {
      const int ch = ch_tile * BCH + wch + a * 16 + lg * 4;
      if (ch >= d.Cout) continue;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (d.bias) {
        const float* bp = d.bias + (long long)n * d.bias_stride + ch;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bp[e];
      }
      const long long o = opix * d.ldo + ch;
      if (d.res_mode == RES_PRE_ACT) {
        float rv[4];
        load4(res + o, rv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act_g(v[e], d.act, d.slope);
      if (d.res_mode == RES_POST_ACT) {
        float rv[4];
        load4(res + o, rv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
      }
      store4(out + o, v);
    }
  }
}

// every K tile inside one tap (see the UNI comment at the 2-stage kernel)
static bool conv_uniform_taps(const ConvDesc& d, int bk) {
  if (d.KD > 8 || d.KH > 8 || d.KW > 8) return false;
  if (d.lcin < 0) return d.ntaps == 1;
  return d.Cin % bk == 0;
}

template <typename T, bool UNI>
static int launch_v3(ConvDesc d, hipStream_t s) {
  constexpr int BCH = 128, BPIX = 256;
  constexpr size_t LDS = 3 * (BCH + BPIX) * 8 * sizeof(uint4);
  d.n_pix_tiles = (int)((d.M + BPIX - 1) / BPIX);
  d.n_ch_tiles = (d.Cout + BCH - 1) / BCH;
  const long long nblk = (long long)d.n_pix_tiles * d.n_ch_tiles;
  RGBM_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  static bool attr_done = false;
  if (!attr_done) {
    RGBM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_v3_kernel<T, UNI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    attr_done = true;
  }
  const int variant = sizeof(T) == 2 ? 13 : 12;
  prof_begin_launch(s, variant, d.algo_flops, d.algo_bytes);
  hipLaunchKernelGGL((conv_igemm_v3_kernel<T, UNI>), dim3((unsigned)nblk), dim3(512), LDS, s, d);
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, int BCH, int BPIX, bool UNI>
static int launch_one_g(ConvDesc d, hipStream_t s) {
  d.n_pix_tiles = (int)((d.M + BPIX - 1) / BPIX);
  d.n_ch_tiles = (d.Cout + BCH - 1) / BCH;
  const long long nblk = (long long)d.n_pix_tiles * d.n_ch_tiles;
  RGBM_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  const int variant = (sizeof(T) == 2 ? 4 : 0) + (BCH == 16 ? 0 : BCH == 32 ? 1 : BCH == 64 ? 2 : 3);
  prof_begin_launch(s, variant, d.algo_flops, d.algo_bytes);
  hipLaunchKernelGGL((conv_igemm_glds_kernel<T, BCH, BPIX, UNI>), dim3((unsigned)nblk), dim3(256), 0, s, d);
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T, bool UNI>
static int launch_t_g(const ConvDesc& d, hipStream_t s) {
  switch (conv_ch_tile(d.Cout)) {
    case 16: return launch_one_g<T, 16, 256, UNI>(d, s);
    case 32: return launch_one_g<T, 32, 256, UNI>(d, s);
    case 64: return launch_one_g<T, 64, 256, UNI>(d, s);
    default: return launch_one_g<T, 128, 128, UNI>(d, s);
  }
}

template <typename T>
static int launch_dtype_g(const ConvDesc& d, hipStream_t s) {
  const bool uni = conv_uniform_taps(d, 8 * (16 / (int)sizeof(T))) && !(g_debug_flags & 16);
  // >= 128 output channels and enough pixel tiles to fill the chip: the 256x128 three-stage kernel
  if (conv_ch_tile(d.Cout) == 128 && !(g_debug_flags & 8) && d.M >= 256 * 256)
    return uni ? launch_v3<T, true>(d, s) : launch_v3<T, false>(d, s);
  return uni ? launch_t_g<T, true>(d, s) : launch_t_g<T, false>(d, s);
}

int launch_conv_glds(const ConvDesc& d, int dtype, hipStream_t s) {
  return dtype == BF16 ? launch_dtype_g<unsigned short>(d, s) : launch_dtype_g<float>(d, s);
}

}  // namespace rgbm
