// Implicit-GEMM convolution, LDS-DMA variant (gfx950): same GEMM view, tiles, fragment maps and epilogue as
// conv_igemm.hip, but both operand tiles travel global -> LDS with `global_load_lds_dwordx4` (no VGPR staging, no
// ds_write pass), so the loads of K tile t+1 are in flight while the MFMAs of tile t run and only one barrier per K tile
// is needed.  The LDS image of a tile is lane-linear (wave-uniform base + lane*16 bytes), therefore the bank-conflict
// swizzle is applied on the SOURCE side: the lane that fills LDS slot (row, j) fetches global chunk j ^ ((row>>1)&7),
// and readers XOR the same value (an involution).  Out-of-image / padded-K lanes fetch from a 16-byte zero page.
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"
#include "prof.h"

#ifndef WS_BUF
#define WS_BUF 1      // request waves of the persistent kernels: LDS-DMA through buffer descriptors (blds16) where the operands fit 32-bit offsets
#endif

namespace rgbm {

extern int g_debug_flags;
extern long long g_ws_min_rows;
// kernel of the 256-channel x 128-pixel launches (layer3 / layer4 / up_1), rgbm_set_tuning("gemm_kernel"): 0 = conv_igemm_ws_kernel
// (16x16x32 MFMAs, 8 multiply waves, 256 x 128 tile), 1 = conv_igemm_m32_kernel<256 x 128>, 2 = conv_igemm_m32_kernel<256 x 256> (default)
int g_gemm_kernel = 2;
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
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_off) : "memory");
}
// one LDS-DMA piece through a buffer descriptor: lane address = base + voff + soff (soff is not part of the range check: a lane whose
// voff is >= num_records reads nothing and delivers zeros).  tools/micro/cu_fill_rate.hip: one wave sustains 37-43 GB/s of 1 KiB pieces
// this way against 27-32 GB/s with 64-bit global addresses, four waves 120-144 against 104-114
__device__ __forceinline__ void blds16(unsigned voff, const __amdgpu_buffer_rsrc_t& rsrc, unsigned soff, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_off) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

// s_barrier with the LDS accesses pinned to their side of it: the builtin alone is IntrNoMem for the compiler, which may hoist a
// later ds_read above it (seen in conv0_sweep.hip's ISA) — a read of a stage another wave has not finished filling
#define RGBM_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <typename T> struct MmaG;
template <> struct MmaG<unsigned short> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4 run2(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct MmaG<f16_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4 run2(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
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

template <> struct MmaG<bx3_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) { c = mma_bx3_k16(a, b, c); }
};
// profiler rows (prof.h) of the type-generic kernels
template <typename T> constexpr int prof_row_generic(int bch) {
  return std::is_same<T, bx3_t>::value ? 26 : (sizeof(T) == 2 ? 4 : 0) + (bch == 16 ? 0 : bch == 32 ? 1 : bch == 64 ? 2 : 3);
}
template <typename T> constexpr int prof_row_ws() { return std::is_same<T, bx3_t>::value ? 29 : sizeof(T) == 2 ? 13 : 12; }

// bx3 K tile (32 channels = the two half-tile chunks of a lane): hi / lo operands of the full-rate 16x16x32 instruction,
// three products per accumulator, small terms first, all (a, b) pairs of a term before the next term
template <int FM, int FN>
__device__ __forceinline__ void mma_bx3_tile(const uint4 (&a0)[FM], const uint4 (&a1)[FM], const uint4 (&b0)[FN], const uint4 (&b1)[FN],
                                             f32x4 (&acc)[FM][FN]) {
  uint4 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
  for (int a = 0; a < FM; ++a) bx3_pair(a0[a], a1[a], ah[a], al[a]);
#pragma unroll
  for (int b = 0; b < FN; ++b) bx3_pair(b0[b], b1[b], bh[b], bl[b]);
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(al[a], bh[b], acc[a][b]);
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(ah[a], bl[b], acc[a][b]);
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(ah[a], bh[b], acc[a][b]);
}

// 8-byte LDS read the compiler cannot merge with its neighbours (and does not count: the caller waits by hand)
template <int OFF>
__device__ __forceinline__ void lds_rd8(uint2& dst, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// the three products of a split-pair tile from operands that are already separated into hi and lo parts
template <int FM, int FN>
__device__ __forceinline__ void mma_bx3_ops(const uint4 (&ah)[FM], const uint4 (&al)[FM], const uint4 (&bh)[FN], const uint4 (&bl)[FN],
                                            f32x4 (&acc)[FM][FN]) {
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(al[a], bh[b], acc[a][b]);
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(ah[a], bl[b], acc[a][b]);
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(ah[a], bh[b], acc[a][b]);
}

template <int N> struct IC { static constexpr int value = N; };

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
        const char* src = ok ? rowp[i] + soff : reinterpret_cast<const char*>(zero);
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
        glds16(wrow[i] + wk, __builtin_amdgcn_readfirstlane(lds_addr(W + (i * 32 + wave * 8) * 8)));
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
    RGBM_BARRIER();
    if (kt + 1 < d.KT) issue(kt + 1, cur ^ 1);
    const uint4* W = lds + cur * STAGE;
    const uint4* X = W + BCH * 8;
    if constexpr (std::is_same<T, bx3_t>::value) {
      uint4 af[2][FM], bf[2][FN];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int cidx = s * 4 + lg;
#pragma unroll
        for (int a = 0; a < FM; ++a) {
          const int row = wch + a * 16 + lr;
          af[s][a] = W[row * 8 + (cidx ^ ((row >> 1) & 7))];
        }
#pragma unroll
        for (int b = 0; b < FN; ++b) {
          const int row = wpix + b * 16 + lr;
          bf[s][b] = X[row * 8 + (cidx ^ ((row >> 1) & 7))];
        }
      }
      mma_bx3_tile<FM, FN>(af[0], af[1], bf[0], bf[1], acc);
    } else {
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
      if constexpr (std::is_same<T, bx3_t>::value) {
        // plain fp32 into the same 4-byte slots: the feature map the plane sweep and the point heads gather from
        if (d.out_f32) { store4(reinterpret_cast<float*>(d.out) + o, v); continue; }
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
        const char* src = ok ? rowp[i] + soff : reinterpret_cast<const char*>(zero);
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
        glds16(wrow[i] + wk, __builtin_amdgcn_readfirstlane(lds_addr(W + (i * 64 + wave * 8) * 8)));
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
    if constexpr (std::is_same<T, bx3_t>::value) {
      uint4 af[2][FM], bf[2][FN];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int cidx = s * 4 + lg;
#pragma unroll
        for (int a = 0; a < FM; ++a) {
          const int row = wch + a * 16 + lr;
          af[s][a] = W[row * 8 + (cidx ^ ((row >> 1) & 7))];
        }
#pragma unroll
        for (int b = 0; b < FN; ++b) {
          const int row = wpix + b * 16 + lr;
          bf[s][b] = X[row * 8 + (cidx ^ ((row >> 1) & 7))];
        }
      }
      mma_bx3_tile<FM, FN>(af[0], af[1], bf[0], bf[1], acc);
      return;
    }
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
          MmaG<T>::run(af[a], bf[b], acc[a][b]);
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
    RGBM_BARRIER();              // every wave's share of tile kt is in LDS; stage of tile kt-1 is free
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

// ---------------------------------------------------------------------------------------------------------
// 256 x 128 tile, role-specialised: 8 multiply waves + 4 request waves, 3-stage LDS ring (uniform-tap layers only).
// Per K tile and wave the 3-stage kernel above issues 97 VALU + 71 SALU + 16 ds_read + 7 VMEM instructions next to its 32
// MFMAs (PMC, MFMA 34 % busy): the request bookkeeping, not bandwidth, sets the pace.  Here the 8 multiply waves execute
// nothing but ds_read + MFMA + one barrier per K tile, and the whole request side (tap walk, validity masks, 12 LDS-DMA
// pieces per wave and tile) lives in 4 extra waves, one per SIMD, whose scalar/vector work runs in the issue gaps of the
// MFMA streams.  Same operand layout, swizzle, fragment maps and epilogue as the kernels above.
// ---------------------------------------------------------------------------------------------------------
// LDS chunk swizzle of the weight rows in conv_igemm_ws_kernel: conflict-free for its permuted fragment rows
__device__ __forceinline__ int swz_w(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

// n / divisor for n < 2^31 with a host-made magic: one 32x32->64 multiply and a shift instead of ~35 instructions
__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned m, int sh) {
  return (unsigned)(((unsigned long long)n * m) >> sh);
}
// output-grid coordinates of GEMM row m
__device__ __forceinline__ void decode_row(const ConvDesc& d, unsigned m, unsigned& n, unsigned& qd, unsigned& qh, unsigned& qw) {
  unsigned t = m;
  unsigned q = fdiv(t, d.fd_m[0], d.fd_s[0]); qw = t - q * (unsigned)d.Wq; t = q;
  q = fdiv(t, d.fd_m[1], d.fd_s[1]); qh = t - q * (unsigned)d.Hq; t = q;
  q = fdiv(t, d.fd_m[2], d.fd_s[2]); qd = t - q * (unsigned)d.Dq; n = q;
}

// WIDE: the tile is 256 channels x 128 pixels instead of 128 x 256 (same LDS, same MFMA count, same 12 pieces per request wave):
// for Cout % 256 == 0 it halves the gathered-pixel bytes that cross L2 -> LDS at the price of twice the weight bytes, and the
// weight slice of a K tile is the same 32 KB for every workgroup of the chip while the pixel rows are a stream from HBM
//
// RH (row halo; with WIDE, 16-bit types; 2-D 3x3, stride 1, pad = dilation <= 4, Cin a multiple of 64): the gathered pixel rows are
// what this kernel pays for — they stream from HBM / MALL through L2 while the weight slice of a K tile is the same for every
// workgroup of the chip (X from the zero page: -29 % time; W from the zero page: -8 %, for equal bytes) — and the three kw taps
// of one kernel row read the same pixels shifted by 0, dil, 2*dil GEMM rows.  So the K order becomes (kh, channel block, kw),
// ONE X slot of 128 + 2*dil rows serves the three steps of a (kh, channel block) group (the multiply waves read their B fragments
// at row offset kw*dil; lanes whose neighbour falls off the image row read a zero row), and only the weights are staged per step:
// 17 X pieces per three steps instead of 48.  Two rings: 3 W slots of 32 KB, 3 X slots of 17 KB (a group's pixels are requested
// over the three steps two groups ahead).
// SLIM: a 64-channel x 256-pixel tile with FOUR multiply waves (8 waves, 512 threads; 10 pieces per request wave and K tile) for
// layers with at most 64 output channels in the storage types that have no ws64 kernel (split pairs, fp32) and for 16-bit
// layers that kernel does not take (residual adds): the role split of this kernel instead of the do-it-all generic tile.
template <typename T, bool WIDE, bool RH, bool SLIM>
__global__ __launch_bounds__(SLIM ? 512 : 768) void conv_igemm_ws_kernel(const ConvDesc d) {
  static_assert(!RH || (WIDE && sizeof(T) == 2), "row-halo variant: 256 x 128 tile, 16-bit storage");
  static_assert(!SLIM || (!WIDE && !RH), "slim tile: plain ring only");
  constexpr int NMW = SLIM ? 4 : 8;          // multiply waves; four request waves follow
  constexpr int NTH = (NMW + 4) * 64;
  constexpr int BCH = SLIM ? 64 : WIDE ? 256 : 128, BPIX = (WIDE && !SLIM) ? 128 : 256;
  constexpr int XROWS = 136;                 // RH: rows of an X slot (BPIX + 2 * 4, whole 8-row pieces)
  constexpr int WSLOT = BCH * 8, XSLOT = XROWS * 8;              // RH: uint4 slots of a W / X ring slot
  constexpr int XBASE = 3 * WSLOT, ZROW = XBASE + 3 * XSLOT;     // RH: X ring, then one row of zeros
  constexpr int E = 16 / sizeof(T);
  constexpr int BK = 8 * E;
  constexpr int XR = BPIX / 32;              // 8 gathered rows per request thread
  constexpr int WL = BCH / 32;               // 4 weight rows per request thread
  constexpr int NP = XR + WL;                // 12 LDS-DMA pieces per request wave and K tile
  constexpr int FM = 4, FN = 4;
  constexpr int STAGE = (BCH + BPIX) * 8;    // uint4 slots per stage (48 KB)
  extern __shared__ __attribute__((aligned(16))) uint4 lds3[];
  static_assert(NP == (SLIM ? 10 : 12), "the counted waits below assume 12 (slim: 10) pieces per tile");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7 multiply, 8..11 request
  const int KT = d.KT;

  // Persistent: workgroup b owns tiles b, b + grid, b + 2*grid, ... (grid is a multiple of 8, so all of them map to the
  // same XCD) and the K-tile ring simply continues across tile boundaries: while the multiply waves write tile t's
  // outputs, the first two K tiles of tile t+1 are already landing.  Measured on the one-tile-per-workgroup version:
  // 15-19 us of prologue + pipeline fill + epilogue + relaunch per tile that nothing overlapped (a quarter of layer4).
  const int n_my = ((int)d.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = n_my * KT;               // K-tile steps of this workgroup; every wave passes `total` barriers
  auto tile_of = [&](int k, int& pix_tile, int& ch_tile) {
    const int v = (int)blockIdx.x + k * (int)gridDim.x;          // virtual workgroup id -> XCD-aware tile order
    const int nblk = d.n_tiles, bq = nblk >> 3, br = nblk & 7, xcd = v & 7, bidx = v >> 3;
    const int lid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
    pix_tile = lid / d.n_ch_tiles;
    ch_tile = lid - pix_tile * d.n_ch_tiles;
  };

  // Per-channel bias table in LDS behind the ring (launch_ws reserves it): the multiply waves must not wait on vmcnt for
  // anything they do not need — on gfx9 the counter also covers their own output stores, in order, so a global bias read
  // after a tile's stores costs a full store round trip (measured: 5 us per tile).
  float* lbias = reinterpret_cast<float*>(lds3 + (RH ? ZROW + 8 : 3 * STAGE));
  const bool bias_lds = d.bias != nullptr && d.bias_stride == 0 && d.Cout <= 2048;
  if (bias_lds)
    for (int i = tid; i < d.Cout; i += NTH) lbias[i] = d.bias[i];
  if (RH && tid < 8) lds3[ZROW + tid] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();

  if constexpr (RH) {
    if (wave >= NMW) {
      // ---------------------------------------------------------------- request waves, row-halo variant
      // W stream: step g+2 (8 pieces per wave and step) behind barrier g; X stream: group g/3 + 2, its 17 pieces dealt
      // round-robin to the four waves (5 / 4 / 4 / 4) and requested over the group's three steps as 2, 2, 1|0 pieces per wave.
      const T* __restrict__ in = reinterpret_cast<const T*>(d.in);
      const T* __restrict__ wgt = reinterpret_cast<const T*>(d.wgt);
      const int pw = wave - NMW;
      const int j = lane & 7, r8 = lane >> 3;
      const int dil = d.dilw;
      const int NCC = d.Cin >> 6;
      const int ngroups = total / 3;
      const char* wrowp[8];
      const char* xrowp[5];
      unsigned xmask[5];
      const char* zero = reinterpret_cast<const char*>(g_zero_page);
      const unsigned ldsb = __builtin_amdgcn_readfirstlane(lds_addr(lds3));
      auto enter_tile_w = [&](int k) {
        int pix_tile, ch_tile;
        tile_of(k, pix_tile, ch_tile);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = (pw + 4 * i) * 8 + r8;
          wrowp[i] = reinterpret_cast<const char*>(wgt + (long long)(ch_tile * BCH + row) * d.Kpad + (j ^ swz_w(row)) * E);
        }
      };
      auto enter_tile_x = [&](int k) {
        int pix_tile, ch_tile;
        tile_of(k, pix_tile, ch_tile);
        const long long p0 = (long long)pix_tile * BPIX;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const int xr = (pw + 4 * i) * 8 + r8;            // LDS row xr holds GEMM row m = p0 - dil + xr (of kernel row kh)
          const long long m = p0 - dil + xr;
          unsigned mk = 0;
          long long pix0 = 0;
          if (pw + 4 * i < XROWS / 8 && xr < BPIX + 2 * dil && m >= 0 && m < d.M) {
            unsigned n, qd, qh, qw;
            decode_row(d, (unsigned)m, n, qd, qh, qw);
            for (int kh = 0; kh < 3; ++kh) mk |= (unsigned)((unsigned)((int)qh + (kh - 1) * dil) < (unsigned)d.Hi) << kh;
            pix0 = ((long long)n * d.Hi + ((int)qh - dil)) * d.Wi + (int)qw;      // input pixel of kernel row 0
          }
          xmask[i] = mk;
          xrowp[i] = reinterpret_cast<const char*>(in) + (pix0 * d.Cin + (j ^ ((xr >> 1) & 7)) * E) * 2ll;
        }
      };
      // W stream state: (kh, cc, kw) of the next step to request, its tile and global step; X stream: (kh, cc) of the group
      int wkh = 0, wcc = 0, wkw = 0, wtile = 0, wg = 0;
      int xkh = 0, xcc = 0, xtile = 0, xg = 0;
      auto issue_w = [&]() {
        if (wg >= total) return 0;
        if (wkh == 0 && wcc == 0 && wkw == 0) enter_tile_w(wtile);
        const unsigned sbase = ldsb + (unsigned)(wg % 3) * (WSLOT * 16);
        const long long woff = ((long long)(wkh * 3 + wkw) * d.Cin + wcc * 64) * 2ll;
#pragma unroll
        for (int i = 0; i < 8; ++i) glds16(wrowp[i] + woff, sbase + ((pw + 4 * i) * 64) * 16);
        if (++wkw == 3) { wkw = 0; if (++wcc == NCC) { wcc = 0; if (++wkh == 3) { wkh = 0; ++wtile; } } }
        ++wg;
        return 8;
      };
      // pieces [i0, i1) of this wave's share of X group xg; part 2 closes the group
      auto issue_x = [&](auto i0c, auto i1c, bool last) {
        constexpr int i0 = decltype(i0c)::value, i1 = decltype(i1c)::value;
        if (xg >= ngroups) return 0;
        if (i0 == 0 && xkh == 0 && xcc == 0) enter_tile_x(xtile);
        const unsigned sbase = ldsb + (unsigned)(XBASE + (xg % 3) * XSLOT) * 16;
        const long long xoff = ((long long)xkh * dil * d.Wi * d.Cin + xcc * 64) * 2ll;
        int n = 0;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
          if (pw + 4 * i < XROWS / 8) {                    // wave-uniform (piece 16 exists for the first wave only)
            const char* src = ((xmask[i] >> xkh) & 1u) ? xrowp[i] + xoff : zero;
            glds16(src, sbase + ((pw + 4 * i) * 64) * 16);
            ++n;
          }
        }
        if (last) { if (++xcc == NCC) { xcc = 0; if (++xkh == 3) { xkh = 0; ++xtile; } } ++xg; }
        return n;
      };
      auto issue_x_all = [&]() { issue_x(IC<0>{}, IC<2>{}, false); issue_x(IC<2>{}, IC<4>{}, false); issue_x(IC<4>{}, IC<5>{}, true); };
      issue_x_all();                          // group 0
      issue_x_all();                          // group 1
      issue_w();                              // step 0
      int after = issue_w();                  // step 1: what is allowed to be in flight behind step 0's operands
      for (int g = 0; g < total; ++g) {
        // everything requested before the last request of step g's weights has landed when at most `after` pieces are in flight
        if (after >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (after == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (after == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RGBM_BARRIER();            // step g's operands are in LDS; the W slot of step g-1 and (every third step) the X slot of group g/3 - 1 are free
        const int ph = g % 3;
        int nx = 0;
        if (ph == 0) nx = issue_x(IC<0>{}, IC<2>{}, false);
        else if (ph == 1) nx = issue_x(IC<2>{}, IC<4>{}, false);
        else nx = issue_x(IC<4>{}, IC<5>{}, true);
        const int nw = issue_w();             // step g+2
        after = nw == 0 ? 0 : nx + nw;        // nothing behind the last weights: drain
      }
      return;
    }
  }

  if (wave >= NMW) {
    // ------------------------------------------------------------------ request waves
#ifdef WS_PRIO
    if (WS_PRIO == 3) __builtin_amdgcn_s_setprio(2);
#endif
    const T* __restrict__ in = reinterpret_cast<const T*>(d.in);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(d.wgt);
    const int pw = wave - NMW;
    const int tp = tid - NMW * 64;
    const int j = tp & 7;
    const int r0 = tp >> 3;                  // 0..31
    const int js = j ^ ((r0 >> 1) & 7);
    const char* rowp[XR];
    unsigned rmask[XR];
    const char* wrow[WL];
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    int tkd = 0, tkh = 0, tkw = 0, tc = 0;   // wave-uniform tap walker
    long long wko = 0;                       // byte offset of the walker's K index in a weight row
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds3 + pw * 64));      // this wave's first piece, stage 0
    // buffer form of the requests (d.buf_ok, set by the launcher when both operands fit 32-bit offsets): X offsets are taken from
    // (input base - xbias) so that a row whose tap (0,0,0) lies in the padding still has a non-negative offset
    const bool use_buf = WS_BUF && d.buf_ok;
    const long long xbias = ((long long)(d.pd * d.Hi + d.ph) * d.Wi + d.pw) * d.Cin * (long long)sizeof(T);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(d.in)) - xbias, 0,
        (int)(unsigned)((long long)d.N * d.Di * d.Hi * d.Wi * d.Cin * (long long)sizeof(T) + xbias), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(d.wgt)), 0, (int)(unsigned)((long long)d.n_ch_tiles * BCH * d.Kpad * (long long)sizeof(T)), 0x00020000);
    unsigned xoff[XR], woff[WL];

    // The eight lanes (r0, j = 0..7) of a row group need the same XR row descriptors (GEMM rows r0 + 32*i): lane j decodes
    // row i = j once and the group exchanges them, instead of every lane decoding all XR rows (3 divisions, 7 bound tests and
    // two 64-bit products each: ~800 instructions per lane and tile, on the critical path of the first request of a tile —
    // with that work skipped a 4-K-tile launch ran 29 % faster, a 36-K-tile layer3 conv 7 %).  The decode of tile k+1 runs right
    // behind the LAST request of tile k (prepare_tile), i.e. while that request flies; the first request of tile k+1 only pays
    // the exchange (enter_tile).
    long long mybase = 0;
    unsigned mymask = 0;
    int my_ch_tile = 0;
    auto prepare_tile = [&](int k) {
      int pix_tile, ch_tile;
      tile_of(k, pix_tile, ch_tile);
      my_ch_tile = ch_tile;
      mybase = 0;
      mymask = 0;
      if (j < XR) {
        const long long m = (long long)pix_tile * BPIX + r0 + 32 * j;
        int xn = 0, xd0 = -(1 << 20), xh0 = 0, xw0 = 0;
        if (m < d.M) {
          unsigned n, qd, qh, qw;
          decode_row(d, (unsigned)m, n, qd, qh, qw);
          xn = (int)n * d.Di;
          xd0 = (int)qd * d.sd - d.pd;
          xh0 = (int)qh * d.sh - d.ph;
          xw0 = (int)qw * d.sw - d.pw;
        }
        for (int kk = 0; kk < d.KD; ++kk) mymask |= (unsigned)((unsigned)(xd0 + kk * d.dild) < (unsigned)d.Di) << kk;
        for (int kk = 0; kk < d.KH; ++kk) mymask |= (unsigned)((unsigned)(xh0 + kk * d.dilh) < (unsigned)d.Hi) << (8 + kk);
        for (int kk = 0; kk < d.KW; ++kk) mymask |= (unsigned)((unsigned)(xw0 + kk * d.dilw) < (unsigned)d.Wi) << (16 + kk);
        const long long pix0 = ((long long)(xn + xd0) * d.Hi + xh0) * d.Wi + xw0;
        mybase = pix0 * d.Cin * (long long)sizeof(T);
      }
    };
    auto enter_tile = [&]() {
      const int grp = (tp & 63) & 56;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const long long b = __shfl(mybase, grp | i, 64);
        rmask[i] = (unsigned)__shfl((int)mymask, grp | i, 64);
        rowp[i] = reinterpret_cast<const char*>(in) + b + (long long)(js * E) * (long long)sizeof(T);
        xoff[i] = (unsigned)(b + xbias + (long long)(js * E) * (long long)sizeof(T));
      }
#pragma unroll
      for (int i = 0; i < WL; ++i) {
        const int row = r0 + 32 * i;
        wrow[i] = reinterpret_cast<const char*>(wgt + (long long)(my_ch_tile * BCH + row) * d.Kpad + (j ^ swz_w(row)) * E);
        woff[i] = (unsigned)(((long long)(my_ch_tile * BCH + row) * d.Kpad + (j ^ swz_w(row)) * E) * (long long)sizeof(T));
      }
      tkd = tkh = tkw = tc = 0;
      wko = 0;
    };

    int ikt = 0, itile = 0;                  // K tile / tile index of the next request
    auto issue = [&](int stage) {
      if (ikt == 0) enter_tile();
      const unsigned sbase = lds0 + (unsigned)stage * (STAGE * 16);
      const unsigned sel = (1u << tkd) | (256u << tkh) | (65536u << tkw);
      const long long soff =
          ((long long)((tkd * d.dild * d.Hi + tkh * d.dilh) * d.Wi + tkw * d.dilw) * d.Cin + tc) * (long long)sizeof(T);
      const bool cok = tkd < d.KD && tc + js * E < d.Cin;
      if (use_buf) {
#pragma unroll
        for (int i = 0; i < XR; ++i) {
          const bool ok = cok && (rmask[i] & sel) == sel;
          blds16(ok ? xoff[i] : 0xffffffffu, rsrc_x, (unsigned)soff, sbase + (BCH * 8 + i * 256) * 16);      // X rows 32*i + 8*pw .. +7
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) blds16(woff[i], rsrc_w, (unsigned)wko, sbase + (i * 256) * 16);       // W rows 32*i + 8*pw .. +7
      } else {
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const bool ok = cok && (rmask[i] & sel) == sel;
        const char* src = ok ? rowp[i] + soff : zero;
        glds16(src, sbase + (BCH * 8 + i * 256) * 16);               // X rows 32*i + 8*pw .. +7
      }
#pragma unroll
      for (int i = 0; i < WL; ++i) glds16(wrow[i] + wko, sbase + (i * 256) * 16);   // W rows 32*i + 8*pw .. +7; wko = byte offset of K index tap * Cin + channel
      }
      if (d.korder && d.lcin >= 0) {
        // channel block outer, taps inner: the taps of one channel block read the same pixel rows shifted by a few rows / columns,
        // so between two uses of a cache line the XCD touches one channel block of its 32 tiles (~0.6 MB) instead of every channel
        // of them (2-4 MB, as much as its L2 holds): the pixels cross the fabric into L2 once per tile instead of once per kernel row
        wko += (long long)d.Cin * (long long)sizeof(T);
        if (++tkw == d.KW) { tkw = 0; if (++tkh == d.KH) { tkh = 0; if (++tkd == d.KD) { tkd = 0; tc += BK; wko = (long long)tc * (long long)sizeof(T); } } }
      } else {
        wko += BK * (long long)sizeof(T);
        tc += BK;
        if (d.lcin >= 0 && tc >= d.Cin) {
          tc = 0;
          if (++tkw == d.KW) { tkw = 0; if (++tkh == d.KH) { tkh = 0; ++tkd; } }
        }
      }
      if (++ikt == KT) { ikt = 0; ++itile; if (itile < n_my) prepare_tile(itile); }
    };

    if (total > 0) prepare_tile(0);
    if (total > 0) issue(0);
    if (total > 1) issue(1);
    int st = 0;
    for (int g = 0; g < total; ++g) {
      if (g + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (SLIM) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // step g landed; step g+1 may still fly
      RGBM_BARRIER();            // step g is complete in LDS; the stage of step g-1 is free
      if (g + 2 < total) issue(st == 0 ? 2 : st - 1);
      st = st == 2 ? 0 : st + 1;
    }
    return;
  }

  // -------------------------------------------------------------------- multiply waves
#ifdef WS_PRIO
  // experiment builds: 1 = the second-dispatched half of the multiply waves at static priority 1, 2 = all multiply waves above
  // the request waves, 3 = request waves above the multiply waves (set in their branch)
  if (WS_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if (WS_PRIO == 2) __builtin_amdgcn_s_setprio(1);
#endif
  const int wch = SLIM ? 0 : WIDE ? (wave >> 1) * 64 : (wave >> 2) * 64;
  const int wpix = (WIDE && !SLIM) ? (wave & 1) * 64 : (wave & 3) * 64;
  const int lr = lane & 15, lg = lane >> 4;
  // Fragment reads run half a K tile ahead of the MFMAs that consume them (two register sets): after the barrier of
  // step g the first-half fragments are requested, the second half of the previous step is multiplied out of registers
  // meanwhile, then the second-half fragments are requested under the first half's MFMAs.
  uint4 af0[FM], bf0[FN], af1[FM], bf1[FN];
  f32x4 acc[FM][FN];
  auto load_half = [&](int st, int s, uint4 (&af)[FM], uint4 (&bf)[FN]) {
    const uint4* W = lds3 + st * STAGE;
    const uint4* X = W + BCH * 8;
    const int cidx = s * 4 + lg;
#pragma unroll
    for (int a = 0; a < FM; ++a) {
      const int row = wch + (lr >> 2) * 16 + a * 4 + (lr & 3);       // MFMA row lr of fragment a = channel (lr>>2)*16 + a*4 + (lr&3)
      af[a] = W[row * 8 + (cidx ^ swz_w(row))];
    }
#pragma unroll
    for (int b = 0; b < FN; ++b) {
      const int row = wpix + b * 16 + lr;
      bf[b] = X[row * 8 + (cidx ^ ((row >> 1) & 7))];
    }
  };
  // RH: W slot ws, X slot xs read at row offset kw * dil; edge lanes (bit b of eLm / eRm) read the zero row
  unsigned eLm = 0, eRm = 0;
  const int rh_dil = RH ? d.dilw : 0;
  auto load_half_rh = [&](int ws, int xs, int kw, int s, uint4 (&af)[FM], uint4 (&bf)[FN]) {
    const uint4* W = lds3 + ws * WSLOT;
    const uint4* X = lds3 + XBASE + xs * XSLOT;
    const int cidx = s * 4 + lg;
#pragma unroll
    for (int a = 0; a < FM; ++a) {
      const int row = wch + (lr >> 2) * 16 + a * 4 + (lr & 3);
      af[a] = W[row * 8 + (cidx ^ swz_w(row))];
    }
    const unsigned em = kw == 0 ? eLm : (kw == 2 ? eRm : 0u);
#pragma unroll
    for (int b = 0; b < FN; ++b) {
      const int row = wpix + b * 16 + lr + kw * rh_dil;
      const uint4* src = ((em >> b) & 1u) ? lds3 + ZROW + cidx : X + row * 8 + (cidx ^ ((row >> 1) & 7));
      bf[b] = *src;
    }
  };
  auto mma_half = [&](const uint4 (&af)[FM], const uint4 (&bf)[FN]) {
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b) MmaG<T>::run(af[a], bf[b], acc[a][b]);
  };
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(d.res);
  // split pairs: byte offsets (inside a stage) of this lane's two chunks of its first A / B fragment row
  unsigned x3a0 = 0, x3a1 = 0, x3b0 = 0, x3b1 = 0;
  if constexpr (std::is_same<T, bx3_t>::value) {
    const int ra = wch + (lr >> 2) * 16 + (lr & 3), rb = wpix + lr;
    x3a0 = (unsigned)(ra * 8 + (lg ^ swz_w(ra))) * 16u;
    x3a1 = (unsigned)(ra * 8 + ((4 + lg) ^ swz_w(ra))) * 16u;
    x3b0 = (unsigned)(BCH * 8 + rb * 8 + (lg ^ ((rb >> 1) & 7))) * 16u;
    x3b1 = (unsigned)(BCH * 8 + rb * 8 + ((4 + lg) ^ ((rb >> 1) & 7))) * 16u;
  }
  int st = 0;
  int xs = 0, kw = 0;                        // RH: X slot and kernel column of the current step
  auto tile_interior = [&](int pix_tile, int ch_tile) {
    return (d.bias == nullptr || bias_lds) && (long long)(pix_tile + 1) * BPIX <= d.M && (ch_tile + 1) * BCH <= d.Cout &&
           d.act != ACT_TANH;
  };
  for (int k = 0; k < n_my; ++k) {
    {
      // interior tile with a per-channel bias (or none): the accumulators start at the bias, and the epilogue below is the
      // straight-line one.  (Tile coordinates are recomputed in the epilogue: nothing of this block lives across the K loop.)
      int pt, ct;
      tile_of(k, pt, ct);
      if (tile_interior(pt, ct) && d.bias) {
        const float* bp = lbias + ct * BCH + wch + lg * 16;
#pragma unroll
        for (int a = 0; a < FM; ++a) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + a * 4);
#pragma unroll
          for (int b = 0; b < FN; ++b) acc[a][b] = b4;
        }
      } else {
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if constexpr (RH) {
      {
        int pt, ct;
        tile_of(k, pt, ct);
        eLm = eRm = 0;
#pragma unroll
        for (int b = 0; b < FN; ++b) {
          const long long m = (long long)pt * BPIX + wpix + b * 16 + lr;
          unsigned n, qd, qh, qw;
          decode_row(d, m < d.M ? (unsigned)m : 0u, n, qd, qh, qw);
          eLm |= (unsigned)((int)qw - rh_dil < 0) << b;
          eRm |= (unsigned)((int)qw + rh_dil >= d.Wi) << b;
        }
      }
      for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RGBM_BARRIER();
        load_half_rh(st, xs, kw, 0, af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt > 0) mma_half(af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        load_half_rh(st, xs, kw, 1, af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        mma_half(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        st = st == 2 ? 0 : st + 1;
        if (++kw == 3) { kw = 0; xs = xs == 2 ? 0 : xs + 1; }
      }
      mma_half(af1, bf1);
    } else if constexpr (std::is_same<T, bx3_t>::value) {
      // split pairs: the two half-tile chunks of a lane (channels 4*lg.. and 16 + 4*lg..) together are the 8 k values of
      // one 16x16x32 operand — hi parts and lo parts separately — so a K tile (32 channels) is 16 x 3 full-rate MFMAs
      // (lo*hi, hi*lo, hi*hi; 16 independent accumulators between two uses of the same one)
      // The hi parts of a lane's two chunks are the first 8 bytes of each, the lo parts the last 8: read as 8-byte halves they land
      // in the operand registers directly.  (Read as two 16-byte chunks, every operand pair cost ~6 v_mov to regroup — 52 per wave
      // and K tile on the MFMA issue port, all behind the full read latency.  The 8-byte reads of a half wave hit each bank pair
      // twice, which costs the LDS array what the 16-byte reads cost.)  The reads are inline asm — hipcc merges plain 8-byte loads
      // back into 16-byte or paired ones and regroups with v_mov again — issued in the order the products need them and waited for
      // with counted waits (LDS reads return in order; at most 16 are outstanding).
      static_assert(!std::is_same<T, bx3_t>::value || (FM == 4 && FN == 4), "counted waits below: 4 x 4 fragments");
#ifdef X3_PAIR_B128      // timing builds (tools/abl_build.sh): the 16-byte reads + regrouping this loop replaced
      for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RGBM_BARRIER();
        load_half(st, 0, af0, bf0);
        load_half(st, 1, af1, bf1);
        mma_bx3_tile<FM, FN>(af0, af1, bf0, bf1, acc);
        st = st == 2 ? 0 : st + 1;
      }
#else
      for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RGBM_BARRIER();
        const unsigned sb = lds_addr(lds3 + st * STAGE);
        const unsigned pa0 = sb + x3a0, pa1 = sb + x3a1, pb0 = sb + x3b0, pb1 = sb + x3b1;
        uint2 ahp[FM][2], alp[FM][2], bhp[FN][2], blp[FN][2];
#define X3_RD_A(dst, a, half) { lds_rd8<(a) * 512 + (half) * 8>(dst[a][0], pa0); lds_rd8<(a) * 512 + (half) * 8>(dst[a][1], pa1); }
#define X3_RD_B(dst, b, half) { lds_rd8<(b) * 2048 + (half) * 8>(dst[b][0], pb0); lds_rd8<(b) * 2048 + (half) * 8>(dst[b][1], pb1); }
#define X3_T(p) "+v"(p[0]), "+v"(p[1])
#ifdef X3_NOWAIT      // timing builds only (wrong results): the products do not wait for their operands = a perfect prefetch
#define X3_WAIT(w) ""
#else
#define X3_WAIT(w) w
#endif
#define X3_OP(p) make_uint4(p[0].x, p[0].y, p[1].x, p[1].y)
        X3_RD_A(alp, 0, 1) X3_RD_B(bhp, 0, 0) X3_RD_B(bhp, 1, 0) X3_RD_B(bhp, 2, 0) X3_RD_B(bhp, 3, 0)
        X3_RD_A(alp, 1, 1) X3_RD_A(alp, 2, 1) X3_RD_A(alp, 3, 1)
        asm volatile(X3_WAIT("s_waitcnt lgkmcnt(6)") : X3_T(alp[0]), X3_T(bhp[0]), X3_T(bhp[1]), X3_T(bhp[2]), X3_T(bhp[3]) :: "memory");
        const uint4 bh[FN] = {X3_OP(bhp[0]), X3_OP(bhp[1]), X3_OP(bhp[2]), X3_OP(bhp[3])};
        {
          const uint4 al = X3_OP(alp[0]);
#pragma unroll
          for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(al, bh[b], acc[0][b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        X3_RD_A(ahp, 0, 0) X3_RD_A(ahp, 1, 0)
        asm volatile(X3_WAIT("s_waitcnt lgkmcnt(8)") : X3_T(alp[1]) :: "memory");
        {
          const uint4 al = X3_OP(alp[1]);
#pragma unroll
          for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(al, bh[b], acc[1][b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        X3_RD_A(ahp, 2, 0) X3_RD_A(ahp, 3, 0)
        asm volatile(X3_WAIT("s_waitcnt lgkmcnt(10)") : X3_T(alp[2]) :: "memory");
        {
          const uint4 al = X3_OP(alp[2]);
#pragma unroll
          for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(al, bh[b], acc[2][b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        X3_RD_B(blp, 0, 1) X3_RD_B(blp, 1, 1)
        asm volatile(X3_WAIT("s_waitcnt lgkmcnt(12)") : X3_T(alp[3]) :: "memory");
        {
          const uint4 al = X3_OP(alp[3]);
#pragma unroll
          for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(al, bh[b], acc[3][b]);
        }
        __builtin_amdgcn_sched_barrier(0);
        X3_RD_B(blp, 2, 1) X3_RD_B(blp, 3, 1)
        asm volatile(X3_WAIT("s_waitcnt lgkmcnt(0)")
                     : X3_T(ahp[0]), X3_T(ahp[1]), X3_T(ahp[2]), X3_T(ahp[3]), X3_T(blp[0]), X3_T(blp[1]), X3_T(blp[2]), X3_T(blp[3])
                     :: "memory");
        const uint4 ah[FM] = {X3_OP(ahp[0]), X3_OP(ahp[1]), X3_OP(ahp[2]), X3_OP(ahp[3])};
        const uint4 bl[FN] = {X3_OP(blp[0]), X3_OP(blp[1]), X3_OP(blp[2]), X3_OP(blp[3])};
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(ah[a], bl[b], acc[a][b]);
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int b = 0; b < FN; ++b) MmaG<unsigned short>::run(ah[a], bh[b], acc[a][b]);
#undef X3_RD_A
#undef X3_RD_B
#undef X3_T
#undef X3_OP
#undef X3_WAIT
        st = st == 2 ? 0 : st + 1;
      }
#endif
    } else {
    for (int kt = 0; kt < KT; ++kt) {
      // every fragment read of the previous step has returned before the request waves may refill its stage
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      RGBM_BARRIER();
      load_half(st, 0, af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt > 0) mma_half(af1, bf1);                  // second half of K tile kt-1
      __builtin_amdgcn_sched_barrier(0);
      load_half(st, 1, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      st = st == 2 ? 0 : st + 1;
    }
    mma_half(af1, bf1);
    }

    // ---- epilogue of tile k (the request waves are already filling the ring for tile k+1) ----
    // All 16 residual reads of a lane are requested before the first one is used: written as load-use-store per
    // fragment they were 16 dependent HBM round trips, 15-17 us per tile with the matrix pipe idle (measured as the
    // K-independent part of the launch time).
    int pix_tile, ch_tile;
    tile_of(k, pix_tile, ch_tile);
    const bool interior = tile_interior(pix_tile, ch_tile);
    long long obase[FN];
    int nn[FN];
    bool pok[FN];
    // every layer but the sub-pixel classes of the transposed convs enumerates its output tensor in storage order: GEMM row m IS
    // output pixel m, and the four coordinate divisions per lane and tile are only needed for a per-sample bias
    const bool dense_out = d.osd == 1 && d.osh == 1 && d.osw == 1 && d.opd == 0 && d.oph == 0 && d.opw == 0 && d.Dq == d.Do &&
                           d.Hq == d.Ho && d.Wq == d.Wo;
#pragma unroll
    for (int b = 0; b < FN; ++b) {
      const long long m = (long long)pix_tile * BPIX + wpix + b * 16 + lr;
      pok[b] = m < d.M;
      if (dense_out && (interior || d.bias_stride == 0)) {
        nn[b] = 0;
        obase[b] = (pok[b] ? m : 0ll) * d.ldo;
        continue;
      }
      unsigned n, qd, qh, qw;
      decode_row(d, pok[b] ? (unsigned)m : 0u, n, qd, qh, qw);
      nn[b] = (int)n;
      obase[b] = ((((long long)n * d.Do + (qd * d.osd + d.opd)) * d.Ho + (qh * d.osh + d.oph)) * d.Wo + (qw * d.osw + d.opw)) * d.ldo;
    }
    // channel map of this kernel: MFMA row r of fragment a is channel (r>>2)*16 + a*4 + (r&3) of the wave's 64, so the four
    // fragments give a lane 16 CONSECUTIVE channels (lg*16 ..) of its pixel: 16-byte residual reads and stores
    // (half as many memory instructions as the plain C layout's 8-byte pieces; the epilogue is issue-bound)
    const int chL = ch_tile * BCH + wch + lg * 16;
    constexpr int CHK = 16 / (int)sizeof(T);             // channels per 16-byte chunk
    constexpr int NQ = 16 / CHK;                         // chunks per lane and pixel
    // Interior tiles with a per-channel bias take a straight-line epilogue specialised on (activation, residual mode):
    // the generic loop below tests pixel / channel bounds, bias, residual mode and activation per 16-byte chunk and
    // reloads the bias in front of every chunk (behind the previous chunk's store, which may alias it) — ~1000
    // instructions, ~100 branches and 8 dependent bias round trips per wave and tile, measured as 9 us per tile with
    // the matrix pipe idle.
    if (interior) {
      // One activation expression for none / ReLU / PReLU: y = v < 0 ? max(v, -FLT_MAX) * nslope : v with nslope = 1 / 0 / slope
      // (the clamp keeps -inf * 0 from becoming NaN; a NaN fails the compare and passes through, like torch).  Specialised on
      // (activation, residual mode) as nine instantiations, hipcc hoisted the 64 compares of a lane in front of the dispatch
      // and kept their masks in 128 SGPRs (spilled to VGPR lanes with s_nop-padded v_writelane / v_readlane).
      const float nslope = d.act == ACT_RELU ? 0.f : d.act == ACT_PRELU ? d.slope : 1.f;
      auto fast = [&](auto resc) {
        constexpr int RES = decltype(resc)::value;
        // the residual reads of TWO pixel fragments are requested before their first store (see the vmcnt note above).  All four
        // at once (32 registers in the 16-bit types) pushed the kernel over its 168 registers: hipcc then spilled an ACCUMULATOR
        // out of the last MFMA block of every tile and reloaded it in the epilogue behind `s_waitcnt vmcnt(0)` — a scratch round
        // trip (~2 us) with the matrix pipe idle, per tile
        constexpr int GRP = 2;
#pragma unroll
        for (int bh = 0; bh < FN; bh += GRP) {
          uint4 rr[NQ][GRP];
          if (RES != RES_NONE) {
#pragma unroll
            for (int bb = 0; bb < GRP; ++bb)
#pragma unroll
              for (int q = 0; q < NQ; ++q)
                rr[q][bb] = *reinterpret_cast<const uint4*>(res + obase[bh + bb] + chL + q * CHK);
          }
#pragma unroll
          for (int bb = 0; bb < GRP; ++bb) {
            const int b = bh + bb;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              float v[CHK], rv[CHK];
#pragma unroll
              for (int e = 0; e < CHK; ++e) v[e] = acc[(q * CHK + e) >> 2][b][(q * CHK + e) & 3];
              if (RES != RES_NONE) unpack_chunk(rr[q][bb], rv, T());
              if (RES == RES_PRE_ACT) {
#pragma unroll
                for (int e = 0; e < CHK; ++e) v[e] += rv[e];
              }
#pragma unroll
              for (int e = 0; e < CHK; ++e) v[e] = v[e] < 0.f ? __builtin_fmaxf(v[e], -3.402823466e38f) * nslope : v[e];
              if (RES == RES_POST_ACT) {
#pragma unroll
                for (int e = 0; e < CHK; ++e) v[e] += rv[e];
              }
              const uint4 pk = pack_chunk(v, T());
              *reinterpret_cast<uint4*>(out + obase[b] + chL + q * CHK) = pk;
            }
          }
        }
      };
      if (d.res_mode == RES_NONE) fast(IC<RES_NONE>{});
      else if (d.res_mode == RES_PRE_ACT) fast(IC<RES_PRE_ACT>{});
      else fast(IC<RES_POST_ACT>{});
      continue;
    }
#pragma unroll
    for (int bh = 0; bh < FN; bh += 2) {                 // two pixel fragments (2*NQ residual reads in flight) at a time
      uint4 rr[NQ][2];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int q = 0; q < NQ; ++q) rr[q][bb] = make_uint4(0u, 0u, 0u, 0u);      // defined on every path: no value carried around the tile loop
      if (d.res_mode != RES_NONE) {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int c = chL + q * CHK;
            const bool ok = pok[bh + bb] && c + CHK <= d.Cout;
            rr[q][bb] = *reinterpret_cast<const uint4*>(res + (ok ? obase[bh + bb] + c : 0ll));      // masked lanes read element 0
          }
      }
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int b = bh + bb;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int c = chL + q * CHK;
          if (!pok[b] || c >= d.Cout) continue;
          const bool whole = c + CHK <= d.Cout;          // Cout is a multiple of 4: a bf16 tail chunk holds 4 channels
          float v[CHK], rv[CHK];
#pragma unroll
          for (int e = 0; e < CHK; ++e) v[e] = acc[(q * CHK + e) >> 2][b][(q * CHK + e) & 3];
          if (d.bias) {
            const float* bp = d.bias + (long long)nn[b] * d.bias_stride + c;
#pragma unroll
            for (int e = 0; e < CHK; ++e) if (e < 4 || whole) v[e] += bp[e];
          }
#pragma unroll
          for (int e = 0; e < CHK; ++e) rv[e] = 0.f;
          if (d.res_mode != RES_NONE) {
            if (whole) unpack_chunk(rr[q][bb], rv, T());
            else load4(res + obase[b] + c, rv);
          }
          if (d.res_mode == RES_PRE_ACT) {
#pragma unroll
            for (int e = 0; e < CHK; ++e) v[e] += rv[e];
          }
#pragma unroll
          for (int e = 0; e < CHK; ++e) v[e] = apply_act_g(v[e], d.act, d.slope);
          if (d.res_mode == RES_POST_ACT) {
#pragma unroll
            for (int e = 0; e < CHK; ++e) v[e] += rv[e];
          }
          if (whole) *reinterpret_cast<uint4*>(out + obase[b] + c) = pack_chunk(v, T());
          else store4(out + obase[b] + c, v);
        }
      }
    }
  }
}

// magic for n / dvs, n < 2^31: q = (n * m) >> sh  (round-up method, sh = 31 + ceil(log2 dvs))
static void make_fastdiv(int dvs, unsigned& m, int& sh) {
  int s = 0;
  while ((1ll << s) < (long long)dvs) ++s;
  sh = 31 + s;
  m = (unsigned)(((1ull << sh) + (unsigned long long)dvs - 1ull) / (unsigned long long)dvs);
}

// both operands addressable with 32-bit byte offsets from one buffer descriptor each (the request waves' blds16 form)
static bool conv_buffer_offsets_ok(const ConvDesc& d, int bch, size_t esz) {
  const long long xbytes = ((long long)d.N * d.Di * d.Hi * d.Wi + (long long)(d.pd * d.Hi + d.ph) * d.Wi + d.pw) * d.Cin * (long long)esz;
  const long long wbytes = (long long)((d.Cout + bch - 1) / bch) * bch * d.Kpad * (long long)esz;
  return xbytes < (1ll << 32) - 65536 && wbytes < (1ll << 32) - 65536 && d.pd >= 0 && d.ph >= 0 && d.pw >= 0;
}

template <typename T, bool WIDE, bool RH, bool SLIM = false>
static int launch_ws(ConvDesc d, hipStream_t s) {
  constexpr int BCH = SLIM ? 64 : WIDE ? 256 : 128, BPIX = (WIDE && !SLIM) ? 128 : 256;
  constexpr size_t LDS = (RH ? (3 * (size_t)(BCH + 136) * 8 + 8) : 3 * (size_t)(BCH + BPIX) * 8) * sizeof(uint4) + 2048 * sizeof(float);      // K-tile ring(s) + per-channel bias table
  d.n_pix_tiles = (int)((d.M + BPIX - 1) / BPIX);
  d.n_ch_tiles = (d.Cout + BCH - 1) / BCH;
  const long long ntiles = (long long)d.n_pix_tiles * d.n_ch_tiles;
  RGBM_REQUIRE(ntiles > 0 && ntiles < (1ll << 31) && d.M < (1ll << 31), "conv grid out of range");
  d.n_tiles = (int)ntiles;
  make_fastdiv(d.Wq, d.fd_m[0], d.fd_s[0]);
  make_fastdiv(d.Hq, d.fd_m[1], d.fd_s[1]);
  make_fastdiv(d.Dq, d.fd_m[2], d.fd_s[2]);
  d.korder = (g_debug_flags & (1 << 20)) ? 0 : 1;          // debug flag 1048576: taps outer, channels inner (the order before round 3) for A/B
  d.buf_ok = conv_buffer_offsets_ok(d, BCH, sizeof(T)) && !(g_debug_flags & (1 << 27));      // debug flag 134217728: 64-bit global addresses + zero page (A/B)
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_igemm_ws_kernel<T, WIDE, RH, SLIM>), (int)LDS)) return rc;
  int n_cu = 0;
  if (int rc = persistent_grid_cus(&n_cu)) return rc;
  const int grid = ntiles < n_cu ? (int)ntiles : n_cu;
  prof_begin_launch(s, SLIM ? (std::is_same<T, bx3_t>::value ? 34 : sizeof(T) == 2 ? 35 : 36) : RH ? 32 : WIDE && sizeof(T) == 2 ? 31 : WIDE && std::is_same<T, bx3_t>::value ? 33 : prof_row_ws<T>(), d.algo_flops, d.algo_bytes);
  hipLaunchKernelGGL((conv_igemm_ws_kernel<T, WIDE, RH, SLIM>), dim3((unsigned)grid), dim3(SLIM ? 512 : 768), LDS, s, d);
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

#include "conv_igemm_m32.inc"      // 256 x 128 tile on 32x32x16 MFMAs, one multiply wave per SIMD (tuning key gemm_kernel)


// ---------------------------------------------------------------------------------------------------------
// 256 pixel x 64 channel tile, bf16, three roles (layers with 33..64 output channels and no residual: up_2, up_3).
// These layers have short K (9..36 K tiles) and 6.6 GB of activations per launch: with the 2-stage kernel the epilogue
// (8-byte stores from the MFMA C layout) and the pipeline refill of every tile left them at 0.43 PFLOP/s / 1.5 TB/s.
//   waves 0-3  multiply: ds_read + MFMA only; at the end of a tile bias + activation + bf16 pack -> LDS staging tile
//   waves 4-7  request : tap walk, validity masks, 10 LDS-DMA pieces per K tile, counted vmcnt (loads only: stores
//                        share vmcnt on gfx9 and complete out of order, so they must not come from these waves)
//   waves 8-11 store   : move the staged tile of the PREVIOUS pixel tile to global memory as whole 128-byte rows, a slice
//                        per K step, while the multiply waves are already deep in the next tile
// Persistent over tiles with one continuous 3-stage K ring, one barrier per K step for all 12 waves plus one final
// barrier that publishes the last staged tile.
// ---------------------------------------------------------------------------------------------------------
//
// RH (row halo; 3x3, stride 1, pad = dilation <= 4, Cin a multiple of 64): a K step is one kernel ROW (kh) of one 64-channel
// input chunk.  The 3 kw taps of that row read the same pixels shifted by 0, dil, 2*dil GEMM rows, so ONE X stage of
// 256 + 2*dil rows serves all three (the multiply waves read their B fragments at row offsets kw*dil; lanes whose
// neighbour falls off the image row read a zero row instead), with the 3 x 64 weight rows of that kernel row beside it.
// Per tile the L2->LDS traffic drops 2.1x (3 x 57 KB instead of 9 x 40 KB for Cin = 64) and the barriers 3x; a step holds
// 96 MFMAs per wave.  Two 57 KB stages (plain double buffering) + the staging tile fit the 160 KB LDS.
template <bool RH, typename T>      // T: unsigned short (bf16) or f16_t
__global__ __launch_bounds__(768) void conv_igemm_ws64_kernel(const ConvDesc d) {
  constexpr int BCH = 64, BPIX = 256;
  constexpr int E = 8, BK = 64;
  constexpr int XR = BPIX / 32;              // 8 gathered rows per request thread
  constexpr int WL = BCH / 32;               // 2 weight rows per request thread
  constexpr int NP = XR + WL;                // 10 pieces per request wave and K tile
  constexpr int FM = 4, FN = 4;
  constexpr int HXROWS = 264;                // RH: X rows per stage (256 + 2*dil, padded to whole 8-row pieces)
  constexpr int HWROWS = 3 * BCH;            // RH: weight rows per stage (kw-major)
  constexpr int NST = RH ? 2 : 3;            // ring depth
  constexpr int STAGE = RH ? (HWROWS + HXROWS) * 8 : (BCH + BPIX) * 8;    // uint4 slots per stage (57 KB / 40 KB)
  constexpr int SROW = 144;                  // bytes per staged pixel row (128 + 16 pad)
  extern __shared__ __attribute__((aligned(16))) uint4 lds3[];
  const uint4* zrow = lds3 + NST * STAGE;    // RH: 8 slots (one row) of zeros
  unsigned char* stg = reinterpret_cast<unsigned char*>(lds3 + NST * STAGE + (RH ? 8 : 0));
  static_assert(NP == 10, "the counted waits below assume 10 pieces per tile");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NCC = d.Cin >> 6;                // RH: 64-channel input chunks
  const int KT = RH ? 3 * NCC : d.KT;        // K steps per tile
  if (RH && tid < 8) lds3[NST * STAGE + tid] = make_uint4(0u, 0u, 0u, 0u);      // visible after the first barrier
  // per-channel bias (zeros if there is none) as an LDS table behind the staging tile: the accumulators start from it, so the
  // multiply waves issue no global load — and wait on no vmcnt — anywhere (see conv_igemm_ws_kernel)
  float* lbias = reinterpret_cast<float*>(stg + BPIX * SROW);
  const bool bias_lds = d.bias == nullptr || d.bias_stride == 0;
  if (tid < BCH) lbias[tid] = (d.bias && bias_lds && tid < d.Cout) ? d.bias[tid] : 0.f;
  __syncthreads();
  const int n_my = ((int)d.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = n_my * KT;               // every wave passes total + 1 barriers
  auto tile_of = [&](int k) {
    const int v = (int)blockIdx.x + k * (int)gridDim.x;
    const int nblk = d.n_tiles, bq = nblk >> 3, br = nblk & 7, xcd = v & 7, bidx = v >> 3;
    return (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;      // pixel tile (one channel tile)
  };

  if (wave >= 8) {
    // ------------------------------------------------------------------ store waves
    const int ts = tid - 512;
    const int c = ts & 7;                     // 16-byte chunk = 8 channels
    const int r0 = ts >> 3;                   // rows r0 + 32*i
    T* __restrict__ out = reinterpret_cast<T*>(d.out);
    const int ch = c * 8;
    const int nch = d.Cout - ch;              // <= 0: nothing, 4: half chunk, >= 8: whole chunk
    const int per = (8 + (KT - 1) - 1) / (KT - 1);       // row groups per K step (KT >= 2)
    auto flush = [&](int k, int i0, int i1) {            // row groups [i0, i1) of tile k
      if (nch <= 0) return;
      const int pix_tile = tile_of(k);
      for (int i = i0; i < i1 && i < 8; ++i) {
        const int r = r0 + 32 * i;
        const long long m = (long long)pix_tile * BPIX + r;
        if (m >= d.M) continue;
        unsigned n, qd, qh, qw;
        decode_row(d, (unsigned)m, n, qd, qh, qw);
        const long long o = ((((long long)n * d.Do + (qd * d.osd + d.opd)) * d.Ho + (qh * d.osh + d.oph)) * d.Wo + (qw * d.osw + d.opw)) * d.ldo + ch;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + r * SROW + c * 16);
        if (nch >= 8) *reinterpret_cast<uint4*>(out + o) = v;
        else *reinterpret_cast<uint2*>(out + o) = make_uint2(v.x, v.y);
      }
    };
    // Fused trailing 1x1 (y2 = act2(W2 . y + bias2), e.g. PSPNet's `final` after up_3): the staged tile already is the B
    // operand layout of an MFMA (pixel row = 64 contiguous channels), so the store waves multiply it by W2 (kept in
    // registers) and write only y2; the 64-channel tensor never reaches HBM.
    const int sw = wave - 8;
    const int lr = lane & 15, lg = lane >> 4;
    uint4 A2[2][2];
    float b2[2][4];
    if (d.w2) {
      const unsigned short* w2 = reinterpret_cast<const unsigned short*>(d.w2);
#pragma unroll
      for (int of = 0; of < 2; ++of)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          A2[of][ks] = make_uint4(0u, 0u, 0u, 0u);
          if (of * 16 < d.cout2) A2[of][ks] = *reinterpret_cast<const uint4*>(w2 + (long long)(of * 16 + lr) * d.kpad2 + ks * 32 + lg * 8);
        }
#pragma unroll
      for (int of = 0; of < 2; ++of)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c2 = of * 16 + lg * 4 + e;
          b2[of][e] = (d.bias2 && c2 < d.cout2) ? d.bias2[c2] : 0.f;
        }
    }
    const int perf = (4 + (KT - 1) - 1) / (KT - 1);       // fused: 16-pixel fragments per K step (4 per wave and tile)
    auto flush_fused = [&](int k, int j0, int j1) {
      const int pix_tile = tile_of(k);
      T* out2 = reinterpret_cast<T*>(d.out2);
      for (int j = j0; j < j1 && j < 4; ++j) {
        const int r = sw * 64 + j * 16 + lr;
        const long long m = (long long)pix_tile * BPIX + r;
        const uint4 B0 = *reinterpret_cast<const uint4*>(stg + r * SROW + lg * 16);
        const uint4 B1 = *reinterpret_cast<const uint4*>(stg + r * SROW + 64 + lg * 16);
        unsigned n, qd, qh, qw;
        decode_row(d, m < d.M ? (unsigned)m : 0u, n, qd, qh, qw);
        const long long o = ((((long long)n * d.Do + (qd * d.osd + d.opd)) * d.Ho + (qh * d.osh + d.oph)) * d.Wo + (qw * d.osw + d.opw)) * d.ldo2;
#pragma unroll
        for (int of = 0; of < 2; ++of) {
          if (of * 16 >= d.cout2) break;
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
          acc = MmaG<T>::run2(A2[of][0], B0, acc);
          acc = MmaG<T>::run2(A2[of][1], B1, acc);
          if (m < d.M) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act_g(acc[e] + b2[of][e], d.act2, d.slope2);
            store4(out2 + o + of * 16 + lg * 4, v);
          }
        }
      }
    };
    int g = 0;
    for (int k = 0; k < n_my; ++k)
      for (int kt = 0; kt < KT; ++kt, ++g) {
        RGBM_BARRIER();
        // the staged tile k-1 was published by barrier (k, 0); it must be drained before barrier (k, KT-1), after which
        // the multiply waves overwrite it
        if (k > 0 && kt < KT - 1) {
          if (d.w2) flush_fused(k - 1, kt * perf, (kt + 1) * perf);
          else flush(k - 1, kt * per, (kt + 1) * per);
        }
      }
    RGBM_BARRIER();             // publishes the last staged tile
    if (d.w2) flush_fused(n_my - 1, 0, 4);
    else flush(n_my - 1, 0, 8);
    return;
  }

  if (wave >= 4) {
    // ------------------------------------------------------------------ request waves
    if constexpr (RH) {
      const T* __restrict__ in = reinterpret_cast<const T*>(d.in);
      const T* __restrict__ wgt = reinterpret_cast<const T*>(d.wgt);
      const int pw = wave - 4;
      const int j = lane & 7, r8 = lane >> 3;
      const int dil = d.dilw;
      constexpr int NXP = HXROWS / 8;                      // 33 X pieces, 24 W pieces per step, dealt round-robin to 4 waves
      constexpr int NWP = HWROWS / 8;
      constexpr int MXP = (NXP + 3) / 4, MWP = NWP / 4;    // per wave: up to 9 X pieces, 6 W pieces
      const char* xrowp[MXP];
      unsigned xmask[MXP];
      const char* wrowp[MWP];
      const char* zero = reinterpret_cast<const char*>(g_zero_page);
      const unsigned ldsb = __builtin_amdgcn_readfirstlane(lds_addr(lds3));
#pragma unroll
      for (int i = 0; i < MWP; ++i) {                      // weight row wr = kw*64 + channel: K index of step (kh, cc) = (kh*3 + kw)*Cin + cc*64
        const int wr = (pw + 4 * i) * 8 + r8;
        const int kw = wr >> 6, ch = wr & 63;
        wrowp[i] = reinterpret_cast<const char*>(wgt + (long long)ch * d.Kpad + (long long)kw * d.Cin + (j ^ ((wr >> 1) & 7)) * E);
      }
      auto enter_tile = [&](int k) {
        const long long p0 = (long long)tile_of(k) * BPIX;
#pragma unroll
        for (int i = 0; i < MXP; ++i) {
          const int xr = (pw + 4 * i) * 8 + r8;            // LDS row xr holds GEMM row m = p0 - dil + xr
          const long long m = p0 - dil + xr;
          unsigned mk = 0;
          long long pix0 = 0;
          if ((pw + 4 * i) < NXP && xr < BPIX + 2 * dil && m >= 0 && m < d.M) {
            unsigned n, qd, qh, qw;
            decode_row(d, (unsigned)m, n, qd, qh, qw);
            for (int kh = 0; kh < 3; ++kh) mk |= (unsigned)((unsigned)((int)qh + (kh - 1) * dil) < (unsigned)d.Hi) << kh;
            pix0 = ((long long)n * d.Hi + ((int)qh - dil)) * d.Wi + (int)qw;      // input pixel of kernel row 0
          }
          xmask[i] = mk;
          xrowp[i] = reinterpret_cast<const char*>(in) + (pix0 * d.Cin + (j ^ ((xr >> 1) & 7)) * E) * 2ll;
        }
      };
      int ikh = 0, icc = 0, itile = 0;
      auto issue = [&](int stage) {
        if (ikh == 0 && icc == 0) enter_tile(itile);
        const unsigned sbase = ldsb + (unsigned)stage * (STAGE * 16);
        const long long xoff = ((long long)ikh * dil * d.Wi * d.Cin + icc * 64) * 2ll;
        const long long woff = ((long long)ikh * 3 * d.Cin + icc * 64) * 2ll;
#pragma unroll
        for (int i = 0; i < MXP; ++i) {
          if (pw + 4 * i < NXP) {                          // wave-uniform
            const char* src = ((xmask[i] >> ikh) & 1u) ? xrowp[i] + xoff : zero;
            glds16(src, sbase + (HWROWS * 8 + (pw + 4 * i) * 64) * 16);
          }
        }
#pragma unroll
        for (int i = 0; i < MWP; ++i) glds16(wrowp[i] + woff, sbase + ((pw + 4 * i) * 64) * 16);
        if (++icc == NCC) { icc = 0; if (++ikh == 3) { ikh = 0; ++itile; } }
      };
      if (total > 0) issue(0);
      for (int g = 0; g < total; ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // step g landed (double buffering: nothing else is in flight)
        RGBM_BARRIER();                        // ... and every wave is done with the other stage
        if (g + 1 < total) issue((g + 1) & 1);
      }
      RGBM_BARRIER();
      return;
    }
    const T* __restrict__ in = reinterpret_cast<const T*>(d.in);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(d.wgt);
    const int pw = wave - 4;
    const int tp = tid - 256;
    const int j = tp & 7;
    const int r0 = tp >> 3;                  // 0..31
    const int js = j ^ ((r0 >> 1) & 7);
    const char* rowp[XR];
    unsigned rmask[XR];
    const char* wrow[WL];
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    int tkd = 0, tkh = 0, tkw = 0, tc = 0;   // wave-uniform tap walker
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds3 + pw * 64));

    // row descriptors: decoded once per 8-lane row group and exchanged, the next tile's behind the last request of the current
    // one (see conv_igemm_ws_kernel)
    long long mybase = 0;
    unsigned mymask = 0;
    auto prepare_tile = [&](int k) {
      const int pix_tile = tile_of(k);
      mybase = 0;
      mymask = 0;
      if (j < XR) {
        const long long m = (long long)pix_tile * BPIX + r0 + 32 * j;
        int xn = 0, xd0 = -(1 << 20), xh0 = 0, xw0 = 0;
        if (m < d.M) {
          unsigned n, qd, qh, qw;
          decode_row(d, (unsigned)m, n, qd, qh, qw);
          xn = (int)n * d.Di;
          xd0 = (int)qd * d.sd - d.pd;
          xh0 = (int)qh * d.sh - d.ph;
          xw0 = (int)qw * d.sw - d.pw;
        }
        for (int kk = 0; kk < d.KD; ++kk) mymask |= (unsigned)((unsigned)(xd0 + kk * d.dild) < (unsigned)d.Di) << kk;
        for (int kk = 0; kk < d.KH; ++kk) mymask |= (unsigned)((unsigned)(xh0 + kk * d.dilh) < (unsigned)d.Hi) << (8 + kk);
        for (int kk = 0; kk < d.KW; ++kk) mymask |= (unsigned)((unsigned)(xw0 + kk * d.dilw) < (unsigned)d.Wi) << (16 + kk);
        const long long pix0 = ((long long)(xn + xd0) * d.Hi + xh0) * d.Wi + xw0;
        mybase = pix0 * d.Cin * (long long)sizeof(T);
      }
    };
    auto enter_tile = [&]() {
      const int grp = (tp & 63) & 56;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const long long bb = __shfl(mybase, grp | i, 64);
        rmask[i] = (unsigned)__shfl((int)mymask, grp | i, 64);
        rowp[i] = reinterpret_cast<const char*>(in) + bb + (long long)(js * E) * (long long)sizeof(T);
      }
      tkd = tkh = tkw = tc = 0;
    };
#pragma unroll
    for (int i = 0; i < WL; ++i)              // one channel tile: the weight rows never change
      wrow[i] = reinterpret_cast<const char*>(wgt + (long long)(r0 + 32 * i) * d.Kpad + js * E);

    int ikt = 0, itile = 0;
    auto issue = [&](int stage) {
      if (ikt == 0) enter_tile();
      const unsigned sbase = lds0 + (unsigned)stage * (STAGE * 16);
      const unsigned sel = (1u << tkd) | (256u << tkh) | (65536u << tkw);
      const long long soff =
          ((long long)((tkd * d.dild * d.Hi + tkh * d.dilh) * d.Wi + tkw * d.dilw) * d.Cin + tc) * (long long)sizeof(T);
      const bool cok = tkd < d.KD && tc + js * E < d.Cin;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const bool ok = cok && (rmask[i] & sel) == sel;
        const char* src = ok ? rowp[i] + soff : zero;
        glds16(src, sbase + (BCH * 8 + i * 256) * 16);
      }
      const long long wk = (long long)ikt * BK * (long long)sizeof(T);
#pragma unroll
      for (int i = 0; i < WL; ++i) glds16(wrow[i] + wk, sbase + (i * 256) * 16);
      tc += BK;
      if (d.lcin >= 0 && tc >= d.Cin) {
        tc = 0;
        if (++tkw == d.KW) { tkw = 0; if (++tkh == d.KH) { tkh = 0; ++tkd; } }
      }
      if (++ikt == KT) { ikt = 0; ++itile; if (itile < n_my) prepare_tile(itile); }
    };

    if (total > 0) prepare_tile(0);
    if (total > 0) issue(0);
    if (total > 1) issue(1);
    int st = 0;
    for (int g = 0; g < total; ++g) {
      if (g + 1 < total) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RGBM_BARRIER();
      if (g + 2 < total) issue(st == 0 ? 2 : st - 1);
      st = st == 2 ? 0 : st + 1;
    }
    RGBM_BARRIER();
    return;
  }

  // -------------------------------------------------------------------- multiply waves
  const int wpix = wave * 64;
  const int lr = lane & 15, lg = lane >> 4;
  uint4 af0[FM], bf0[FN], af1[FM], bf1[FN];
  f32x4 acc[FM][FN];
  auto load_half = [&](int st, int s, uint4 (&af)[FM], uint4 (&bf)[FN]) {
    const uint4* W = lds3 + st * STAGE;
    const uint4* X = W + BCH * 8;
    const int cidx = s * 4 + lg;
#pragma unroll
    for (int a = 0; a < FM; ++a) {
      const int row = a * 16 + lr;
      af[a] = W[row * 8 + (cidx ^ ((row >> 1) & 7))];
    }
#pragma unroll
    for (int b = 0; b < FN; ++b) {
      const int row = wpix + b * 16 + lr;
      bf[b] = X[row * 8 + (cidx ^ ((row >> 1) & 7))];
    }
  };
  auto mma_half = [&](const uint4 (&af)[FM], const uint4 (&bf)[FN]) {
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int b = 0; b < FN; ++b) MmaG<T>::run(af[a], bf[b], acc[a][b]);
  };
  int st = 0;
  for (int k = 0; k < n_my; ++k) {
#pragma unroll
    for (int a = 0; a < FM; ++a) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbias + a * 16 + lg * 4);      // zeros unless bias_lds with a bias
#pragma unroll
      for (int b = 0; b < FN; ++b) acc[a][b] = b4;
    }
    if constexpr (RH) {
      // per tile: which lanes lose their left / right neighbour to the image edge (they read the zero row instead)
      const int dil = d.dilw;
      const long long p0 = (long long)tile_of(k) * BPIX;
      bool eL[FN], eR[FN];
#pragma unroll
      for (int b = 0; b < FN; ++b) {
        const long long m = p0 + wpix + b * 16 + lr;
        unsigned n, qd, qh, qw;
        decode_row(d, m < d.M ? (unsigned)m : 0u, n, qd, qh, qw);
        eL[b] = (int)qw - dil < 0;
        eR[b] = (int)qw + dil >= d.Wi;
      }
      for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RGBM_BARRIER();
        const uint4* W = lds3 + (st & 1) * STAGE;
        const uint4* X = W + HWROWS * 8;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int cidx = s2 * 4 + lg;
            uint4 af[FM], bf[FN];
#pragma unroll
            for (int a = 0; a < FM; ++a) {
              const int row = kw * 64 + a * 16 + lr;
              af[a] = W[row * 8 + (cidx ^ ((row >> 1) & 7))];
            }
#pragma unroll
            for (int b = 0; b < FN; ++b) {
              const int row = wpix + b * 16 + lr + kw * dil;
              const bool edge = kw == 0 ? eL[b] : (kw == 2 ? eR[b] : false);
              const uint4* src = edge ? zrow + cidx : X + row * 8 + (cidx ^ ((row >> 1) & 7));
              bf[b] = *src;
            }
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
              for (int b = 0; b < FN; ++b) MmaG<T>::run(af[a], bf[b], acc[a][b]);
          }
        }
        ++st;
      }
    } else {
    for (int kt = 0; kt < KT; ++kt) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragment reads returned, staged tile written
      RGBM_BARRIER();
      load_half(st, 0, af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt > 0) mma_half(af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      load_half(st, 1, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      st = st == 2 ? 0 : st + 1;
    }
    mma_half(af1, bf1);
    }
    // ---- bias + activation + bf16 pack into the staging tile (the store waves have drained the previous one) ----
    if (bias_lds && d.act != ACT_TANH) {
      // the bias is already in the accumulators; activation fixed at compile time per variant: straight-line code
      const float slope = d.slope;
      auto pack_tile = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int b = 0; b < FN; ++b) {
          const int row = wpix + b * 16 + lr;
#pragma unroll
          for (int a = 0; a < FM; ++a) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x = acc[a][b][e];
              v[e] = ACT == ACT_RELU ? (x < 0.f ? 0.f : x) : ACT == ACT_PRELU ? (x < 0.f ? x * slope : x) : x;
            }
            store4(reinterpret_cast<T*>(stg + row * SROW + (a * 16 + lg * 4) * 2), v);
          }
        }
      };
      if (d.act == ACT_RELU) pack_tile(IC<ACT_RELU>{});
      else if (d.act == ACT_PRELU) pack_tile(IC<ACT_PRELU>{});
      else pack_tile(IC<ACT_NONE>{});
      continue;
    }
    const int pix_tile = tile_of(k);
#pragma unroll
    for (int b = 0; b < FN; ++b) {
      const int row = wpix + b * 16 + lr;
      int n = 0;
      if (d.bias && d.bias_stride != 0) {
        const long long m = (long long)pix_tile * BPIX + row;
        unsigned nn, qd, qh, qw;
        decode_row(d, m < d.M ? (unsigned)m : 0u, nn, qd, qh, qw);
        n = (int)nn;
      }
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        const int ch = a * 16 + lg * 4;
        float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
        if (d.bias && !bias_lds && ch < d.Cout) {
          const float* bp = d.bias + (long long)n * d.bias_stride + ch;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bp[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act_g(v[e], d.act, d.slope);
        store4(reinterpret_cast<T*>(stg + row * SROW + ch * 2), v);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  RGBM_BARRIER();
}

// row-halo variant of the 64-channel kernel: 2-D 3x3, stride 1, "same" padding, Cin a multiple of 64
static bool conv_rowhalo_ok(const ConvDesc& d) {
  return d.KD == 1 && d.KH == 3 && d.KW == 3 && d.sd == 1 && d.sh == 1 && d.sw == 1 && d.Dq == 1 && d.Di == 1 &&
         d.dilh == d.dilw && d.dilw >= 1 && d.dilw <= 4 && d.ph == d.dilh && d.pw == d.dilw && d.pd == 0 &&
         d.lcin >= 6 && d.Hq == d.Hi && d.Wq == d.Wi && d.Wi > 2 * d.dilw && d.osh == 1 && d.osw == 1 && !(g_debug_flags & 256);
}

template <typename T>
static int launch_ws64(ConvDesc d, hipStream_t s) {
  constexpr int BCH = 64, BPIX = 256;
  const bool rh = conv_rowhalo_ok(d);
  const size_t LDS = (rh ? 2 * (size_t)(3 * BCH + 264) * 8 * sizeof(uint4) + 128 + (size_t)BPIX * 144
                         : 3 * (size_t)(BCH + BPIX) * 8 * sizeof(uint4) + (size_t)BPIX * 144) + BCH * sizeof(float);   // + bias table
  d.n_pix_tiles = (int)((d.M + BPIX - 1) / BPIX);
  d.n_ch_tiles = 1;
  d.n_tiles = d.n_pix_tiles;
  make_fastdiv(d.Wq, d.fd_m[0], d.fd_s[0]);
  make_fastdiv(d.Hq, d.fd_m[1], d.fd_s[1]);
  make_fastdiv(d.Dq, d.fd_m[2], d.fd_s[2]);
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_igemm_ws64_kernel<false, T>), 160 * 1024)) return rc;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_igemm_ws64_kernel<true, T>), 160 * 1024)) return rc;
  int n_cu = 0;
  if (int rc = persistent_grid_cus(&n_cu)) return rc;
  const int grid = d.n_tiles < n_cu ? d.n_tiles : n_cu;
  prof_begin_launch(s, 15, d.algo_flops, d.algo_bytes);
  if (rh) hipLaunchKernelGGL((conv_igemm_ws64_kernel<true, T>), dim3((unsigned)grid), dim3(768), LDS, s, d);
  else hipLaunchKernelGGL((conv_igemm_ws64_kernel<false, T>), dim3((unsigned)grid), dim3(768), LDS, s, d);
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// GEMM rows (output pixels) from which the persistent role-specialised kernels replace the generic tiles.  Rounds 1-3 used 65 536
// (one 256-pixel tile per CU).  Round 4, forward + post-processing latency at small batches on one box: with the 64 x 256 tile taken
// for launches that fit one round of the grid (launch_dtype_g) the persistent kernels win from ~1000 rows on in every 16-bit and
// split-pair case — bf16 B = 1 (1568 rows in layer3) 1.86 -> 1.54 ms, B = 8 2.65 -> 2.43 ms; split pairs B = 1 3.35 -> 2.51 ms, B = 2
// 3.56 -> 2.83, B = 4 4.02 -> 3.36, B = 8 5.18 -> 4.62 ms — and fp32 does not care (7.1 / 8.3 / 11.0 / 17.6 ms either way).
static long long ws_min_rows(size_t /*elem_bytes*/) { return g_ws_min_rows > 0 ? g_ws_min_rows : 1024; }

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
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_igemm_v3_kernel<T, UNI>), (int)LDS)) return rc;
  prof_begin_launch(s, prof_row_ws<T>(), d.algo_flops, d.algo_bytes);
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
  prof_begin_launch(s, prof_row_generic<T>(BCH), d.algo_flops, d.algo_bytes);
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
  if (d.out_f32) {      // only the generic tile's epilogue knows the plain-fp32 output form
    RGBM_REQUIRE((std::is_same<T, bx3_t>::value) && d.w2 == nullptr, "out_f32 is a bf16x3 option of the generic kernel");
    return uni ? launch_t_g<T, true>(d, s) : launch_t_g<T, false>(d, s);
  }
  // >= 128 output channels and enough pixel tiles to fill the chip: the 256x128 three-stage kernel
  if (conv_ch_tile(d.Cout) == 128 && !(g_debug_flags & 8) && d.M >= ws_min_rows(sizeof(T))) {
    // role-specialised (uniform taps, 16-byte aligned output / residual rows)
    const unsigned long long al = (unsigned long long)d.out | ((unsigned long long)d.ldo * sizeof(T)) | (d.res ? (unsigned long long)d.res : 0ull);
    if (uni && !(g_debug_flags & 64) && (al & 15ull) == 0ull) {
      // Small launches (B = 1 .. 8): the 256 x 128 / 128 x 256 tiles leave most CUs idle (layer3 at B = 1: 13 tiles), and every tile walks
      // the whole K range.  The 64-channel x 256-pixel shape of the same kernel makes 2-4x as many tiles of a quarter / half of the work;
      // taken while even those fit one round of the persistent grid (debug flag 16777216: never).
      if (!(g_debug_flags & (1 << 24)) && d.Cout % 64 == 0 && d.M < (1ll << 31)) {
        int n_cu = 0;
        if (int rc = persistent_grid_cus(&n_cu)) return rc;
        const long long slim_tiles = ((d.M + 255) / 256) * (d.Cout / 64);
        if constexpr (sizeof(T) == 2 || std::is_same<T, bx3_t>::value) {
          // round 6: launches that do not fill the grid with 256-pixel tiles of the 256-channel layers pick their tile by requested bytes
          // (m32_small_choice; debug flag 1073741824: as in round 5)
          if (d.Cout % 256 == 0 && ((d.M + 255) / 256) * (d.Cout / 256) < n_cu && g_gemm_kernel >= 1 && !(g_debug_flags & ((1 << 30) | 65536)) &&
              conv_buffer_offsets_ok(d, 256, sizeof(T))) {
            // (gemm_kernel = 1, the A/B reference: the same arithmetic on 256-channel x 128-pixel tiles - bit-identical results)
            const int pick = g_gemm_kernel == 1 ? 256 : m32_small_choice(d, n_cu, slim_tiles <= n_cu);
            if (pick) return launch_m32_small<T>(d, s, pick);
          }
          // layer2's 128-channel layers at one to four poses (13-52 tiles of 64 x 256): 64-channel x 128-pixel tiles of the 32x32x16 kernel with
          // their K loop split (conv_igemm_m32.inc) — taken only where the split applies, i.e. while twice the tiles still fit the grid
          if (d.Cout == 128 && ((d.M + 127) / 128) * 2 * 2 <= n_cu && g_gemm_kernel == 2 && !(g_debug_flags & ((1 << 30) | 65536 | 16384 | 32768)) &&
              d.w2 == nullptr && conv_buffer_offsets_ok(d, 64, sizeof(T)))
            return launch_m32_small<T>(d, s, 64);
        }
        if (slim_tiles <= n_cu) return launch_ws<T, false, false, true>(d, s);
      }
      if (d.Cout % 256 == 0 && !(g_debug_flags & 65536)) {
        if constexpr (sizeof(T) == 2 || std::is_same<T, bx3_t>::value) {
          if (conv_buffer_offsets_ok(d, 256, sizeof(T))) {      // its request waves address both operands through 32-bit buffer offsets
            if (g_gemm_kernel >= 1) return launch_m32<T>(d, s, g_gemm_kernel);
          }
        }
        return launch_ws<T, true, false>(d, s);
      }
      // (round 6 measured layer2's 128-channel layers on 128 x 256 tiles of the 32x32x16 kernel: 0.122 against 0.119 ms per launch in bf16,
      // 0.293 against 0.291 in split pairs - a 128-channel tile needs 48 KB per 1024 cycles of MFMA and is request-bound in either kernel)
      return launch_ws<T, false, false>(d, s);
    }
    return uni ? launch_v3<T, true>(d, s) : launch_v3<T, false>(d, s);
  }
  // 33..64 output channels, bf16, no residual, >= 2 K tiles, 16-byte aligned output rows: three-role persistent kernel
  // 33..64 output channels, 16-bit storage: the three-role persistent kernel
  if constexpr (sizeof(T) == 2) {
    if (conv_ws64_eligible(d, BF16)) return launch_ws64<T>(d, s);
  }
  RGBM_REQUIRE(d.w2 == nullptr, "a fused 1x1 needs the ws64 kernel (check conv_ws64_eligible first)");
  // 33..64 output channels where there is no ws64 kernel (split pairs, fp32) or it does not apply (residual adds): the
  // role-specialised kernel with a 64 x 256 tile and four multiply waves
  if (conv_ch_tile(d.Cout) == 64 && uni && d.M >= ws_min_rows(sizeof(T)) && d.M < (1ll << 31) && !(g_debug_flags & (8 | 64 | 262144))) {
    const unsigned long long al = (unsigned long long)d.out | ((unsigned long long)d.ldo * sizeof(T)) | (d.res ? (unsigned long long)d.res : 0ull);
    // (round 6 measured 64-channel x 256-pixel tiles of the 32x32x16 kernel for layer1's 64-channel layers at batch 256: split pairs 0.542 ms
    // per launch against 0.475 on this tile, bf16 38.39 ms per forward against 38.10 with the row-halo ws64 kernel — not dispatched)
    if ((al & 15ull) == 0ull) return launch_ws<T, false, false, true>(d, s);
  }
  return uni ? launch_t_g<T, true>(d, s) : launch_t_g<T, false>(d, s);
}

bool conv_ws64_eligible(const ConvDesc& d, int dtype) {
  if ((dtype != BF16 && dtype != F16) || (g_debug_flags & (4 | 16 | 128))) return false;
  if (!conv_uniform_taps(d, 64) || conv_ch_tile(d.Cout) != 64 || d.res_mode != RES_NONE || d.KT < 2) return false;
  if (d.M < ws_min_rows(2) || d.M >= (1ll << 31)) return false;
  if (d.w2) return d.Cout == 64 && d.kpad2 == 64 && (d.cout2 == 16 || d.cout2 == 32) && d.ldo2 % 4 == 0;
  return (((unsigned long long)d.out | ((unsigned long long)d.ldo * 2ull)) & 15ull) == 0ull;
}

int launch_conv_glds(const ConvDesc& d, int dtype, hipStream_t s) {
  return dtype == BF16 ? launch_dtype_g<unsigned short>(d, s) : dtype == F16 ? launch_dtype_g<f16_t>(d, s)
         : dtype == BF16X3 ? launch_dtype_g<bx3_t>(d, s) : launch_dtype_g<float>(d, s);
}

}  // namespace rgbm
