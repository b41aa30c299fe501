// Halo-tiled 3-D convolution for the thin-channel cost-regularisation stack (gfx950 / CDNA4).
//
// Replaces nn.Conv3d / nn.ConvTranspose3d (+ folded BatchNorm3d + ReLU + skip add) of CostRegNet
//   /root/reference/models/pose_estimator/AdaPose/lib/network_v5.py:260-291
// and, in WARP mode, also homo_warping + the fused-volume add that feed conv0
//   /root/reference/models/pose_estimator/AdaPose/lib/network_v5.py:378-430
// so the 32x24x224x224 plane-sweep volume (154 MB fp32 per view in the reference) is never written to HBM.
//
// Why not the generic implicit GEMM: with 8..64 channels an im2col-style gather re-reads every input voxel
// 27 times from L2 for a handful of MFMAs (measured 62 TFLOP/s).  Here a workgroup stages the input halo of
// its output tile ONCE into LDS (one 16-byte-padded row per voxel, conflict-free for ds_read_b128), then each
// wave walks the taps: B operand (16 voxels x 32 k) straight out of the halo with a per-tap offset, A operand
// (16 output channels x 32 k, BN folded, pre-swizzled per lane on the host) from L1/L2, MFMA 16x16x32 bf16
// (or 16x16x4 f32) accumulating in registers.  Transposed convs run their 8 sub-pixel classes off one halo.
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace rgbm {

template <typename T> struct Mma3;
template <> struct Mma3<unsigned short> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma3<float> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

template <typename T, int CIN, int COUTP, int TD, int TH, int TW, int STRIDE, bool TR>
struct C3Cfg {
  static constexpr int E = 16 / sizeof(T);
  static constexpr int BPT = CIN * sizeof(T);          // bytes per tap (= per voxel)
  static constexpr int VS = BPT + 16;                  // LDS bytes per halo voxel (padded)
  static constexpr int CPV = BPT / 16;                 // 16-byte chunks per voxel
  static constexpr int KE = TR ? 2 : 3;                // taps per axis covered by the halo
  static constexpr int HD = (TD - 1) * STRIDE + KE, HH = (TH - 1) * STRIDE + KE, HW = (TW - 1) * STRIDE + KE;
  static constexpr int NVH = HD * HH * HW;
  static constexpr int LDS_BYTES = (NVH + 1) * VS;     // +1: an all-zero voxel for padded taps
  static constexpr int NV = TD * TH * TW;
  static constexpr int NF = NV / 64;                   // 16-voxel fragments per wave
  static constexpr int FM = COUTP / 16;
  static constexpr int SPT = BPT >= 64 ? BPT / 64 : 1; // MFMA steps per tap
  static constexpr int TPS = BPT >= 64 ? 1 : 64 / BPT; // taps per MFMA step
  static_assert(NV % 64 == 0, "tile must hold a multiple of 64 voxels");
  static_assert(BPT % 16 == 0, "voxel must be a multiple of 16 bytes");
};

__device__ __forceinline__ void warp_ixy(const float* __restrict__ hm, float x, float y, float depth, int H, int W, float& ix,
                                         float& iy) {
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

template <typename T, int CIN, int COUTP, int TD, int TH, int TW, int STRIDE, bool TR, bool WARP>
__global__ __launch_bounds__(256) void conv3d_tile_kernel(const Conv3dTileDesc d) {
  using Cfg = C3Cfg<T, CIN, COUTP, TD, TH, TW, STRIDE, TR>;
  constexpr int VS = Cfg::VS, CPV = Cfg::CPV, HD = Cfg::HD, HH = Cfg::HH, HW = Cfg::HW, NVH = Cfg::NVH;
  constexpr int NF = Cfg::NF, FM = Cfg::FM, SPT = Cfg::SPT, TPS = Cfg::TPS, E = Cfg::E;
  constexpr int NPASS = TR ? 8 : 1;
  constexpr int MAXSTEPS = TR ? (27 * SPT + 8) : ((27 + TPS - 1) / TPS * SPT);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* stepoff = reinterpret_cast<int*>(smem);                       // [MAXSTEPS][4] byte offsets into the halo
  unsigned char* halo = smem + ((MAXSTEPS * 16 + 15) / 16) * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;

  // ---- tile coordinates (q-grid = output grid for conv, input grid for transposed) ----
  int t = blockIdx.x;
  const int tw = t % d.ntw; t /= d.ntw;
  const int th = t % d.nth; t /= d.nth;
  const int td = t % d.ntd; t /= d.ntd;
  const int n = t;
  const int q0d = td * TD, q0h = th * TH, q0w = tw * TW;
  const int i0d = TR ? q0d : q0d * STRIDE - 1, i0h = TR ? q0h : q0h * STRIDE - 1, i0w = TR ? q0w : q0w * STRIDE - 1;

  // ---- per-(step, lane-group) halo byte offsets; padded taps point at the zero voxel ----
  for (int i = tid; i < MAXSTEPS * 4; i += 256) {
    const int s = i >> 2, g = i & 3;
    int off = -1;                                                    // padded tap: read the zero voxel (absolute)
    if (!TR) {
      const int tap = SPT > 1 ? s / SPT : s * TPS + g / (4 / TPS);
      const int chunk = SPT > 1 ? (s % SPT) * 4 + g : g % (4 / TPS);
      if (tap < 27) { const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3; off = ((kd * HH + kh) * HW + kw) * VS + chunk * 16; }
    } else {
      // passes (classes) are concatenated; class c = (pd,ph,pw) has (1+pd)(1+ph)(1+pw) taps
      int rem = s, cls = 0, nst = 0;
      for (cls = 0; cls < 8; ++cls) {
        const int nt = (1 + ((cls >> 2) & 1)) * (1 + ((cls >> 1) & 1)) * (1 + (cls & 1));
        nst = (nt + TPS - 1) / TPS * SPT;
        if (rem < nst) break;
        rem -= nst;
      }
      if (cls < 8) {
        const int KH = 1 + ((cls >> 1) & 1), KW = 1 + (cls & 1);
        const int nt = (1 + ((cls >> 2) & 1)) * KH * KW;
        const int tap = SPT > 1 ? rem / SPT : rem * TPS + g / (4 / TPS);
        const int chunk = SPT > 1 ? (rem % SPT) * 4 + g : g % (4 / TPS);
        if (tap < nt) { const int kd = tap / (KH * KW), kh = (tap / KW) % KH, kw = tap % KW; off = ((kd * HH + kh) * HW + kw) * VS + chunk * 16; }
      }
    }
    stepoff[i] = off;
  }
  if (tid < VS / 16) reinterpret_cast<uint4*>(halo + NVH * VS)[tid] = make_uint4(0u, 0u, 0u, 0u);

  // ---- stage the input halo ----
  {
    const T* __restrict__ in = reinterpret_cast<const T*>(d.in);
    constexpr int TOTAL = NVH * CPV;
    for (int idx = tid; idx < TOTAL; idx += 256) {
      const int vox = idx / CPV, chunk = idx - vox * CPV;
      const int hw = vox % HW, hh = (vox / HW) % HH, hd = vox / (HW * HH);
      const int gd = i0d + hd, gh = i0h + hh, gw = i0w + hw;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if ((unsigned)gd < (unsigned)d.Di && (unsigned)gh < (unsigned)d.Hi && (unsigned)gw < (unsigned)d.Wi) {
        if (!WARP) {
          v = *reinterpret_cast<const uint4*>(in + ((((long long)n * d.Di + gd) * d.Hi + gh) * d.Wi + gw) * CIN + chunk * E);
        } else {
          // fused plane-sweep volume: feat[v] + bilinear(feat[partner], homography(v, depth gd, pixel (gw, gh)))
          const int vv = d.v0 + n, partner = (vv + d.B) % d.V, b = vv % d.B;
          const T* __restrict__ feat = reinterpret_cast<const T*>(d.feat);
          float ix, iy;
          warp_ixy(d.homog + (long long)vv * 12, (float)gw, (float)gh, d.depths[b * d.Di + gd], d.Hi, d.Wi, ix, iy);
          float r[E], acc[E], s[E];
          unpack_chunk(*reinterpret_cast<const uint4*>(feat + (((long long)vv * d.Hi + gh) * d.Wi + gw) * CIN + chunk * E), r, T());
          if (!(isfinite(ix) && isfinite(iy))) {
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = __builtin_nanf("");
          } else {
            ix = fminf(fmaxf(ix, -4.f), 1.0e6f);
            iy = fminf(fmaxf(iy, -4.f), 1.0e6f);
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float tx = ix - fx, ty = iy - fy;
            const T* src = feat + (long long)partner * d.Hi * d.Wi * CIN + chunk * E;
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = 0.f;
            const bool xin0 = (unsigned)x0 < (unsigned)d.Wi, xin1 = (unsigned)(x0 + 1) < (unsigned)d.Wi;
            const bool yin0 = (unsigned)y0 < (unsigned)d.Hi, yin1 = (unsigned)(y0 + 1) < (unsigned)d.Hi;
            if (xin0 && yin0) { unpack_chunk(*reinterpret_cast<const uint4*>(src + ((long long)y0 * d.Wi + x0) * CIN), s, T());
              const float w = (1.f - tx) * (1.f - ty);
#pragma unroll
              for (int e = 0; e < E; ++e) acc[e] += s[e] * w; }
            if (xin1 && yin0) { unpack_chunk(*reinterpret_cast<const uint4*>(src + ((long long)y0 * d.Wi + x0 + 1) * CIN), s, T());
              const float w = tx * (1.f - ty);
#pragma unroll
              for (int e = 0; e < E; ++e) acc[e] += s[e] * w; }
            if (xin0 && yin1) { unpack_chunk(*reinterpret_cast<const uint4*>(src + ((long long)(y0 + 1) * d.Wi + x0) * CIN), s, T());
              const float w = (1.f - tx) * ty;
#pragma unroll
              for (int e = 0; e < E; ++e) acc[e] += s[e] * w; }
            if (xin1 && yin1) { unpack_chunk(*reinterpret_cast<const uint4*>(src + ((long long)(y0 + 1) * d.Wi + x0 + 1) * CIN), s, T());
              const float w = tx * ty;
#pragma unroll
              for (int e = 0; e < E; ++e) acc[e] += s[e] * w; }
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = r[e] + acc[e];
          }
          v = pack_chunk(acc, T());
        }
      }
      *reinterpret_cast<uint4*>(halo + vox * VS + chunk * 16) = v;
    }
  }
  __syncthreads();

  // ---- per-lane fragment bases: voxel (lane&15) of fragment f, tap (0,0,0) ----
  int base[NF];
  int qd_[NF], qh_[NF], qw_[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int vt = (wave * NF + f) * 16 + lr;          // voxel index inside the tile, w fastest
    const int w_ = vt % TW, h_ = (vt / TW) % TH, d_ = vt / (TW * TH);
    base[f] = ((d_ * STRIDE * HH + h_ * STRIDE) * HW + w_ * STRIDE) * VS;
    qd_[f] = q0d + d_; qh_[f] = q0h + h_; qw_[f] = q0w + w_;
  }

  const unsigned char* __restrict__ wg = reinterpret_cast<const unsigned char*>(d.wgt);
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(d.res);
  int step0 = 0;
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    int nsteps;
    if (!TR) nsteps = (27 + TPS - 1) / TPS * SPT;
    else nsteps = ((1 + ((pass >> 2) & 1)) * (1 + ((pass >> 1) & 1)) * (1 + (pass & 1)) + TPS - 1) / TPS * SPT;
    f32x4 acc[FM][NF];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A fragments: [step][COUTP rows][64 bytes]; lane reads row (fm*16+lr), 16-byte chunk lg
    const unsigned char* wp = wg + ((long long)step0 * COUTP + lr) * 64 + lg * 16;
    uint4 an[FM];
#pragma unroll
    for (int a = 0; a < FM; ++a) an[a] = *reinterpret_cast<const uint4*>(wp + a * 16 * 64);
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
      uint4 ac[FM];
#pragma unroll
      for (int a = 0; a < FM; ++a) ac[a] = an[a];
      if (s + 1 < nsteps) {
        const unsigned char* wq = wp + (long long)(s + 1) * COUTP * 64;
#pragma unroll
        for (int a = 0; a < FM; ++a) an[a] = *reinterpret_cast<const uint4*>(wq + a * 16 * 64);
      }
      const int so = stepoff[(step0 + s) * 4 + lg];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        // padded taps (so < 0) must hit the zero voxel itself, not zero-voxel + pixel offset: their weights are 0,
        // but 0 * (stale NaN bytes beyond the halo) would still poison the accumulator
        const uint4 b = *reinterpret_cast<const uint4*>(halo + (so < 0 ? NVH * VS : base[f] + so));
#pragma unroll
        for (int a = 0; a < FM; ++a) Mma3<T>::run(ac[a], b, acc[a][f]);
      }
    }
    step0 += nsteps;

    // ---- epilogue: bias (folded BN), ReLU, post-activation skip add, 4-channel vector store ----
    const int pd = TR ? (pass >> 2) & 1 : 0, ph = TR ? (pass >> 1) & 1 : 0, pw = TR ? pass & 1 : 0;
    constexpr int OS = TR ? 2 : 1;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (qd_[f] >= d.Dq || qh_[f] >= d.Hq || qw_[f] >= d.Wq) continue;
      const long long opix = (((long long)n * d.Do + (qd_[f] * OS + pd)) * d.Ho + (qh_[f] * OS + ph)) * d.Wo + (qw_[f] * OS + pw);
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        const int ch = a * 16 + lg * 4;
        if (ch >= d.Cout) continue;
        float v[4] = {acc[a][f][0], acc[a][f][1], acc[a][f][2], acc[a][f][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += d.bias[ch + e];
          if (d.relu) v[e] = v[e] < 0.f ? 0.f : v[e];      // NaN propagates, like torch.relu
        }
        const long long o = opix * d.Cout + ch;
        if (d.res) {
          float rv[4];
          load4(res + o, rv);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rv[e];
        }
        store4(out + o, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// Number of MFMA steps of one pass (ntaps taps) for a given bytes-per-tap.
static inline int c3_steps(int ntaps, int bpt) { return bpt >= 64 ? ntaps * (bpt / 64) : (ntaps + 64 / bpt - 1) / (64 / bpt); }

// Pack weights into the per-lane A-fragment order the kernel reads: [pass][step][COUTP][4 groups][E elems].
// w: conv [Cout][Cin][27]; transposed [Cin][Cout][27].  Returns fp32 values (converted to the dtype by the caller).
void conv3d_tile_pack(const float* w, const float* scale, int Cin, int Cout, int coutp, bool transposed, int dtype,
                      std::vector<float>& packed) {
  const int E = dtype == BF16 ? 8 : 4;
  const int bpt = Cin * (int)dtype_size(dtype);
  const int spt = bpt >= 64 ? bpt / 64 : 1, tps = bpt >= 64 ? 1 : 64 / bpt;
  const int npass = transposed ? 8 : 1;
  int total = 0;
  for (int p = 0; p < npass; ++p) {
    const int nt = transposed ? (1 + ((p >> 2) & 1)) * (1 + ((p >> 1) & 1)) * (1 + (p & 1)) : 27;
    total += c3_steps(nt, bpt);
  }
  packed.assign((size_t)total * coutp * 4 * E, 0.f);
  int step0 = 0;
  for (int p = 0; p < npass; ++p) {
    const int pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
    const int KD = transposed ? 1 + pd : 3, KH = transposed ? 1 + ph : 3, KW = transposed ? 1 + pw : 3;
    const int nt = KD * KH * KW;
    const int ns = c3_steps(nt, bpt);
    for (int s = 0; s < ns; ++s)
      for (int g = 0; g < 4; ++g) {
        const int tap = spt > 1 ? s / spt : s * tps + g / (4 / tps);
        const int chunk = spt > 1 ? (s % spt) * 4 + g : g % (4 / tps);
        if (tap >= nt) continue;
        const int kd = tap / (KH * KW), kh = (tap / KW) % KH, kw = tap % KW;
        int widx;
        if (!transposed) widx = kd * 9 + kh * 3 + kw;
        else {
          auto kidx = [](int par, int delta) { return par == 0 ? 1 : (delta == 0 ? 2 : 0); };
          widx = kidx(pd, kd) * 9 + kidx(ph, kh) * 3 + kidx(pw, kw);
        }
        for (int o = 0; o < Cout; ++o)
          for (int e = 0; e < E; ++e) {
            const int c = chunk * E + e;
            if (c >= Cin) continue;
            const float val = transposed ? w[((long long)c * Cout + o) * 27 + widx] : w[((long long)o * Cin + c) * 27 + widx];
            packed[(((size_t)(step0 + s) * coutp + o) * 4 + g) * E + e] = val * (scale ? scale[o] : 1.f);
          }
      }
    step0 += ns;
  }
}

template <typename T, int CIN, int COUTP, int TD, int TH, int TW, int STRIDE, bool TR, bool WARP>
static int launch_c3(Conv3dTileDesc d, hipStream_t s) {
  using Cfg = C3Cfg<T, CIN, COUTP, TD, TH, TW, STRIDE, TR>;
  constexpr int NPASSSTEPS = TR ? (27 * Cfg::SPT + 8) : ((27 + Cfg::TPS - 1) / Cfg::TPS * Cfg::SPT);
  constexpr size_t LDS = ((NPASSSTEPS * 16 + 15) / 16) * 16 + Cfg::LDS_BYTES;
  static_assert(LDS <= 160 * 1024, "tile does not fit LDS");
  d.ntd = (d.Dq + TD - 1) / TD; d.nth = (d.Hq + TH - 1) / TH; d.ntw = (d.Wq + TW - 1) / TW;
  const long long nblk = (long long)d.N * d.ntd * d.nth * d.ntw;
  RGBM_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv3d grid out of range");
  auto kern = conv3d_tile_kernel<T, CIN, COUTP, TD, TH, TW, STRIDE, TR, WARP>;
  static bool attr_done = false;
  if (!attr_done) {
    RGBM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    attr_done = true;
  }
  prof_begin_launch(s, d.prof_variant, d.algo_flops, d.algo_bytes);
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), LDS, s, d);
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// layer ids: 0..6 = conv0..conv6, 7..9 = conv7/9/11 (transposed), 10 = conv0 with fused warp
int launch_conv3d_tile(int layer, int dtype, const Conv3dTileDesc& d, hipStream_t s) {
#define C3_CASE(L, CIN, COUTP, TDB, THB, TWB, TDF, THF, TWF, STRIDE, TR, WARP)                                     \
  case L:                                                                                                          \
    return dtype == BF16 ? launch_c3<unsigned short, CIN, COUTP, TDB, THB, TWB, STRIDE, TR, WARP>(d, s)            \
                         : launch_c3<float, CIN, COUTP, TDF, THF, TWF, STRIDE, TR, WARP>(d, s);
  switch (layer) {
    //        layer cin coutp  bf16 tile   f32 tile   stride tr    warp
    C3_CASE(0, 32, 16, 6, 8, 8, 4, 8, 8, 1, false, false)
    C3_CASE(10, 32, 16, 6, 8, 8, 4, 8, 8, 1, false, true)
    C3_CASE(1, 8, 16, 2, 8, 8, 2, 8, 8, 2, false, false)
    C3_CASE(2, 16, 16, 4, 8, 8, 4, 8, 8, 1, false, false)
    C3_CASE(3, 16, 32, 2, 8, 8, 2, 8, 8, 2, false, false)
    C3_CASE(4, 32, 32, 2, 8, 8, 2, 8, 8, 1, false, false)
    C3_CASE(5, 32, 64, 1, 8, 8, 1, 8, 8, 2, false, false)
    C3_CASE(6, 64, 64, 1, 8, 8, 1, 8, 8, 1, false, false)
    C3_CASE(7, 64, 32, 1, 8, 8, 1, 8, 8, 1, true, false)
    C3_CASE(8, 32, 16, 2, 8, 8, 2, 8, 8, 1, true, false)
    C3_CASE(9, 16, 16, 4, 8, 8, 4, 8, 8, 1, true, false)
    default: break;
  }
#undef C3_CASE
  set_error("conv3d_tile: unknown layer id");
  return -1;
}

}  // namespace rgbm
