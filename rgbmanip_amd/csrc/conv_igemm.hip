// Generic implicit-GEMM convolution for gfx950 (CDNA4): 1-D/2-D/3-D, stride, dilation,
// sub-pixel (transposed-conv) output mapping, fused bias / residual / activation epilogue.
//
// Replaces the cuDNN conv calls behind the reference's nn.Conv2d / nn.Conv3d /
// nn.ConvTranspose3d / nn.Conv1d(k=1) / nn.Linear layers
// (/root/reference/models/pose_estimator/AdaPose/lib/pspnet.py:11-30,97-107,
//  lib/network_v5.py:8-28,217-291,317-376).
//
// GEMM view (per workgroup):   D[ch][pix] = sum_k W[ch][k] * X[k][pix]
//   A operand = weights  [BCH rows ][BK]  (K contiguous; pre-packed, BN folded, zero padded)
//   B operand = gathered [BPIX rows][BK]  input patch (NDHWC, k = tap*Cin + c)
// so every lane ends up with 4 consecutive output channels of one pixel (MFMA 16x16 C layout:
// col = lane&15 -> pixel, row = (lane>>4)*4+r -> channel) and stores them as one 8/16-byte write.
//
// 256 threads = 4 wave64; wave tile = min(BCH,64) channels x 64 pixels; K tile = 128 bytes per row
// (64 bf16 / 32 f32), register-staged global->LDS double buffering, one barrier per K tile.
// LDS rows are 128 B with a 16-B-chunk XOR swizzle (chunk ^= (row>>1)&7) so that the 16 rows a
// ds_read_b128 lane group touches land on 16 distinct 16-B bank slots.
// bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate); f32: v_mfma_f32_16x16x4_f32 (exact fp32).
#include <type_traits>
#include "common.h"
#include "prof.h"

namespace rgbm {

extern int g_debug_flags;
int launch_conv_glds(const ConvDesc& d, int dtype, hipStream_t s);

#ifdef RGBM_EXPERIMENTS      // the register-staged kernel (debug flag 4): superseded by the LDS-DMA kernels, kept for A/B
template <typename T> struct Mma;
template <> struct Mma<unsigned short> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    // lane-group g, element e of the 16-byte chunk is k = g*4+e; the same map is used for A and B,
    // so the four MFMAs together contract the 16 k values this chunk quad holds.
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

template <> struct Mma<bx3_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) { c = mma_bx3_k16(a, b, c); }
};

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  // NaN must propagate like torch.relu / prelu (a degenerate pair ends as default_bbox, never as a finite box)
  if (act == ACT_RELU) return v < 0.f ? 0.f : v;
  if (act == ACT_PRELU) return v < 0.f ? v * slope : v;
  if (act == ACT_TANH) return tanhf(v);
  return v;
}

template <typename T, int BCH, int BPIX>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvDesc d) {
  constexpr int E = 16 / sizeof(T);      // elements per 16-byte chunk
  constexpr int BK = 8 * E;              // K tile (128 bytes per row)
  constexpr int XR = BPIX / 32;          // gathered rows per thread
  constexpr int WL = BCH >= 32 ? BCH / 32 : 1;
  constexpr int WCH = BCH < 64 ? BCH : 64;
  constexpr int FM = WCH / 16;
  constexpr int FN = 4;
  constexpr int STAGE = (BCH + BPIX) * 8;   // uint4 slots per stage
  __shared__ uint4 lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // XCD-aware tile order: the hardware dispatches block b to XCD b%8; give each XCD a contiguous
  // run of logical tiles (bijective for any grid size) so channel tiles of one pixel tile and
  // neighbouring pixel tiles (shared halos) hit the same L2.
  const int nblk = gridDim.x;
  const int bq = nblk >> 3, br = nblk & 7;
  const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  const int lid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
  const int pix_tile = lid / d.n_ch_tiles;
  const int ch_tile = lid - pix_tile * d.n_ch_tiles;

  const T* __restrict__ in = reinterpret_cast<const T*>(d.in);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(d.wgt);

  // ---- per-thread gather rows ----------------------------------------------------------
  const int j = tid & 7;        // 16-byte chunk within the 128-byte K tile row
  const int r0 = tid >> 3;      // 0..31
  int xn[XR], xd0[XR], xh0[XR], xw0[XR];
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
  }
  const float rcp_khw = 1.0f / (float)(d.KH * d.KW);
  const float rcp_kw = 1.0f / (float)d.KW;
  const int khw = d.KH * d.KW;

  uint4 xr[XR], wr[WL];
  const bool wload = (BCH >= 32) || (tid < BCH * 8);

  auto gload = [&](int kt) {
    const int k = kt * BK + j * E;
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
      const bool ok = tapok && (unsigned)dd < (unsigned)d.Di && (unsigned)hh < (unsigned)d.Hi &&
                      (unsigned)ww < (unsigned)d.Wi;
      const long long pix = ((long long)(xn[i] + dd) * d.Hi + hh) * d.Wi + ww;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) v = *reinterpret_cast<const uint4*>(in + pix * d.Cin + c);
      xr[i] = v;
    }
    if (wload) {
#pragma unroll
      for (int i = 0; i < WL; ++i) {
        const int row = ch_tile * BCH + r0 + 32 * i;
        wr[i] = *reinterpret_cast<const uint4*>(wgt + (long long)row * d.Kpad + k);
      }
    }
  };
  auto lstore = [&](int stage) {
    uint4* W = lds + stage * STAGE;
    uint4* X = W + BCH * 8;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int row = r0 + 32 * i;
      X[row * 8 + (j ^ ((row >> 1) & 7))] = xr[i];
    }
    if (wload) {
#pragma unroll
      for (int i = 0; i < WL; ++i) {
        const int row = r0 + 32 * i;
        W[row * 8 + (j ^ ((row >> 1) & 7))] = wr[i];
      }
    }
  };

  // ---- wave tile -------------------------------------------------------------------------
  const int wch = (BCH == 128) ? (wave >> 1) * 64 : 0;
  const int wpix = (BCH == 128) ? (wave & 1) * 64 : wave * 64;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lg = lane >> 4;

  gload(0);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < d.KT; ++kt) {
    const bool more = kt + 1 < d.KT;
    if (more) gload(kt + 1);
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
        for (int b = 0; b < FN; ++b) Mma<T>::run(af[a], bf[b], acc[a][b]);
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: bias, residual, activation, 4-channel vector store ---------------------------
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
      for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], d.act, d.slope);
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

#endif  // RGBM_EXPERIMENTS

int conv_ch_tile(int Cout) {
  if (Cout <= 16) return 16;
  if (Cout <= 32) return 32;
  if (Cout <= 64) return 64;
  return 128;
}
int conv_bk(int dtype) { return 8 * dtype_chunk(dtype); }      // 128 bytes per row

#ifdef RGBM_EXPERIMENTS
template <typename T, int BCH, int BPIX>
static int launch_one(ConvDesc d, hipStream_t s) {
  d.n_pix_tiles = (int)((d.M + BPIX - 1) / BPIX);
  d.n_ch_tiles = (d.Cout + BCH - 1) / BCH;
  const long long nblk = (long long)d.n_pix_tiles * d.n_ch_tiles;
  RGBM_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv grid out of range");
  const int variant = std::is_same<T, bx3_t>::value ? 26 : (sizeof(T) == 2 ? 4 : 0) + (BCH == 16 ? 0 : BCH == 32 ? 1 : BCH == 64 ? 2 : 3);
  prof_begin_launch(s, variant, d.algo_flops, d.algo_bytes);
  hipLaunchKernelGGL((conv_igemm_kernel<T, BCH, BPIX>), dim3((unsigned)nblk), dim3(256), 0, s, d);
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename T>
static int launch_t(const ConvDesc& d, hipStream_t s) {
  switch (conv_ch_tile(d.Cout)) {
    case 16: return launch_one<T, 16, 256>(d, s);
    case 32: return launch_one<T, 32, 256>(d, s);
    case 64: return launch_one<T, 64, 256>(d, s);
    default: return launch_one<T, 128, 128>(d, s);
  }
}

#endif  // RGBM_EXPERIMENTS

int launch_conv(const ConvDesc& d, int dtype, hipStream_t s) {
  RGBM_REQUIRE(d.M > 0 && d.M < (1ll << 31), "conv M out of range");
  RGBM_REQUIRE(d.Cout % 4 == 0 && d.ldo % 4 == 0, "conv Cout/ldo must be multiples of 4");
  RGBM_REQUIRE(d.KT > 0 && d.Kpad == d.KT * conv_bk(dtype), "conv K padding mismatch");
  const int E = dtype_chunk(dtype);
  RGBM_REQUIRE(d.Cin % E == 0, "conv Cin must be a multiple of the 16-byte chunk");
  if (d.lcin >= 0) {
    RGBM_REQUIRE((1 << d.lcin) == d.Cin, "conv lcin mismatch");
  } else {
    RGBM_REQUIRE(d.ntaps == 1, "linear-K mode needs a single tap");
  }
#ifdef RGBM_EXPERIMENTS
  if (g_debug_flags & 4)
    return dtype == BF16 ? launch_t<unsigned short>(d, s) : dtype == F16 ? launch_t<f16_t>(d, s)
           : dtype == BF16X3 ? launch_t<bx3_t>(d, s) : launch_t<float>(d, s);
#endif
  return launch_conv_glds(d, dtype, s);      // the LDS-DMA kernels (conv_igemm_glds.hip)
}

}  // namespace rgbm
