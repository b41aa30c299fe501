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
// wave walks the taps fully unrolled: B operand (16 voxels x 32 k) straight out of the halo at a compile-time
// offset, A operand (16 output channels x 32 k, BN folded, pre-swizzled per lane on the host) from L1/L2,
// MFMA 16x16x32 bf16 (or 16x16x4 f32) accumulating in registers.  Transposed convs run their 8 sub-pixel
// classes off one halo.  All loads of the staging phase are issued in batches so that many are in flight.
#include <type_traits>
#include <utility>
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
template <> struct Mma3<f16_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
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

template <> struct Mma3<bx3_t> {      // split pairs (common.h): one chunk = 4 k values, hi*hi + lo*hi + hi*lo
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) { c = mma_bx3_k16(a, b, c); }
};

// WC: the four waves of a workgroup split as (4 / WC fragment groups) x (WC slices of the output channels).  A wave with all
// output channels (WC = 1) loads FM packed weight operands per step for NF B operands; with WC > 1 it loads FM / WC for NF * WC —
// the weights (every wave streams them from L2 for every tile) weigh less, the LDS-resident halo is read more often.
template <typename T, int CIN, int COUTP, int TD, int TH, int TW, int STRIDE, bool TR, int WC = 1>
struct C3Cfg {
  static constexpr int E = 16 / sizeof(T);
  static constexpr int BPT = CIN * sizeof(T);          // bytes per tap (= per voxel)
  // LDS bytes per halo voxel: padded by 16 (conflict-free b128 rows), except 16-byte voxels (conv1): unpadded the halo is
  // 23 KB instead of 46 KB, 6 workgroups per CU instead of 3 — this layer is bound by bytes in flight (3.5 -> 2.9 ms),
  // the 2-way bank conflict on its few MFMA operand reads is not
  static constexpr int VS = BPT <= 32 ? BPT : BPT + 16;
  static constexpr int CPV = BPT / 16;                 // 16-byte chunks per voxel
  static constexpr int KE = TR ? 2 : 3;                // taps per axis covered by the halo
  static constexpr int HD = (TD - 1) * STRIDE + KE, HH = (TH - 1) * STRIDE + KE, HW = (TW - 1) * STRIDE + KE;
  static constexpr int NVH = HD * HH * HW;
  static constexpr int LDS_BYTES = (NVH + 1) * VS;     // +1: an all-zero voxel for padded taps
  static constexpr int NV = TD * TH * TW;
  static constexpr int NF = NV / 64 * WC;              // 16-voxel fragments per wave
  static constexpr int FM = COUTP / 16 / WC;           // 16-channel row tiles per wave
  static_assert(4 % WC == 0 && (COUTP / 16) % WC == 0, "the channel tiles must split over WC waves");
  static constexpr int SPT = BPT >= 64 ? BPT / 64 : 1; // MFMA steps per tap
  static constexpr int TPS = BPT >= 64 ? 1 : 64 / BPT; // taps per MFMA step
  static constexpr int GPT = 4 / TPS;                  // lane groups per tap inside one step
  static_assert(NV % 64 == 0, "tile must hold a multiple of 64 voxels");
  static_assert(BPT % 16 == 0, "voxel must be a multiple of 16 bytes");
  // pass p of a transposed conv = sub-pixel class (pd,ph,pw) with (1+pd)(1+ph)(1+pw) taps; a plain conv has one 27-tap pass
  static constexpr int kd_of(int p) { return TR ? 1 + ((p >> 2) & 1) : 3; }
  static constexpr int kh_of(int p) { return TR ? 1 + ((p >> 1) & 1) : 3; }
  static constexpr int kw_of(int p) { return TR ? 1 + (p & 1) : 3; }
  static constexpr int taps_of(int p) { return kd_of(p) * kh_of(p) * kw_of(p); }
  static constexpr int steps_of(int p) { return (taps_of(p) + TPS - 1) / TPS * SPT; }
  static constexpr int step0_of(int p) { int s = 0; for (int q = 0; q < p; ++q) s += steps_of(q); return s; }
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

int g_debug_flags = 0;
long long g_ws_min_rows = 0;              // GEMM rows from which the persistent role-specialised implicit-GEMM kernels are used; 0 = per storage type (conv_igemm_glds.hip: ws_min_rows), set by rgbm_set_tuning
int g_tuning_version = 0;                 // bumped by rgbm_debug_flags / rgbm_set_tuning: captured forward graphs of older settings are dropped

template <int N> struct IC { static constexpr int value = N; };
// compile-time loop: f(IC<0>{}), ..., f(IC<N-1>{}) — register arrays indexed by the loop variable stay registers
template <int... I, typename F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(IC<I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <typename T, int CIN, int COUTP, int TD, int TH, int TW, int STRIDE, bool TR, bool WARP, int WC = 1>
__global__ __launch_bounds__(256, 2) void conv3d_tile_kernel(const Conv3dTileDesc d) {
  using Cfg = C3Cfg<T, CIN, COUTP, TD, TH, TW, STRIDE, TR, WC>;
  constexpr int VS = Cfg::VS, CPV = Cfg::CPV, HH = Cfg::HH, HW = Cfg::HW, NVH = Cfg::NVH;
  constexpr int NF = Cfg::NF, FM = Cfg::FM, SPT = Cfg::SPT, TPS = Cfg::TPS, GPT = Cfg::GPT, E = Cfg::E;
  constexpr int ZERO_OFF = NVH * VS;
  extern __shared__ __attribute__((aligned(16))) unsigned char halo[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;

  // ---- tile coordinates (q-grid = output grid for conv, input grid for transposed) ----
  // XCD-aware order: hardware block b runs on XCD b%8; give every XCD a contiguous run of tiles (whole views) so the
  // feature maps / halos its 32 CUs gather from stay in that XCD's 4 MB L2 (bijective for any grid size)
  const int nblk = gridDim.x, bq = nblk >> 3, br = nblk & 7, xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  int t = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
  const int tw = t % d.ntw; t /= d.ntw;
  const int th = t % d.nth; t /= d.nth;
  const int td = t % d.ntd; t /= d.ntd;
  const int n = t;
  if (d.tile_mask && !d.tile_mask[(long long)n * (d.tile_mask_stride ? d.tile_mask_stride : d.nth * d.ntw) + th * d.ntw + tw]) return;      // nothing downstream reads this output tile
  const int q0d = td * TD, q0h = th * TH, q0w = tw * TW;
  const int i0d = TR ? q0d : q0d * STRIDE - 1, i0h = TR ? q0h : q0h * STRIDE - 1, i0w = TR ? q0w : q0w * STRIDE - 1;

  if (tid < VS / 16) reinterpret_cast<uint4*>(halo + ZERO_OFF)[tid] = make_uint4(0u, 0u, 0u, 0u);

  // ---- stage the input halo (all global loads of a batch are issued before the first LDS write) ----
  if constexpr (!WARP) {
    const T* __restrict__ in = reinterpret_cast<const T*>(d.in);
    constexpr int TOTAL = NVH * CPV;
    constexpr int ITERS = (TOTAL + 255) / 256;
    constexpr int BATCH = 8;
#pragma unroll 1
    for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
      uint4 v[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = tid + (it0 + u) * 256;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (it0 + u < ITERS && idx < TOTAL) {
          const int vox = idx / CPV, chunk = idx - vox * CPV;
          const int hw = vox % HW, hh = (vox / HW) % HH, hd = vox / (HW * HH);
          const int gd = i0d + hd, gh = i0h + hh, gw = i0w + hw;
          if ((unsigned)gd < (unsigned)d.Di && (unsigned)gh < (unsigned)d.Hi && (unsigned)gw < (unsigned)d.Wi)
            v[u] = *reinterpret_cast<const uint4*>(in + ((((long long)n * d.Di + gd) * d.Hi + gh) * d.Wi + gw) * CIN + chunk * E);
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = tid + (it0 + u) * 256;
        if (it0 + u < ITERS && idx < TOTAL) {
          const int vox = idx / CPV, chunk = idx - vox * CPV;
          *reinterpret_cast<uint4*>(halo + vox * VS + chunk * 16) = v[u];
        }
      }
    }
  } else {
    // fused plane-sweep volume: voxel = feat[v] + bilinear(feat[partner], homography(v, depth, pixel)); one task per
    // halo voxel, 4 chunks at a time: 4 reference + 16 corner loads in flight per lane.
    const T* __restrict__ feat = reinterpret_cast<const T*>(d.feat);
    const int vv = d.v0 + n, partner = (vv + d.B) % d.V, bb = vv % d.B;
    const float* hm = d.homog + (long long)vv * 12;
    const T* refb = feat + (long long)vv * d.Hi * d.Wi * CIN;
    const T* srcb = feat + (long long)partner * d.Hi * d.Wi * CIN;
    constexpr int ITERS = (NVH + 255) / 256;
    constexpr int CG = CPV < 4 ? CPV : 4;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
      const int vox = tid + it * 256;
      if (vox >= NVH) break;
      const int hw = vox % HW, hh = (vox / HW) % HH, hd = vox / (HW * HH);
      const int gd = i0d + hd, gh = i0h + hh, gw = i0w + hw;
      const bool inb = (unsigned)gd < (unsigned)d.Di && (unsigned)gh < (unsigned)d.Hi && (unsigned)gw < (unsigned)d.Wi;
      unsigned char* dst = halo + vox * VS;
      if (!inb) {
#pragma unroll
        for (int c = 0; c < CPV; ++c) *reinterpret_cast<uint4*>(dst + c * 16) = make_uint4(0u, 0u, 0u, 0u);
        continue;
      }
      float ix, iy;
      warp_ixy(hm, (float)gw, (float)gh, d.depths[bb * d.Di + gd], d.Hi, d.Wi, ix, iy);
      const bool fin = isfinite(ix) && isfinite(iy);
      ix = fin ? fminf(fmaxf(ix, -4.f), 1.0e6f) : 0.f;
      iy = fin ? fminf(fmaxf(iy, -4.f), 1.0e6f) : 0.f;
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float tx = ix - fx, ty = iy - fy;
      const bool xin0 = (unsigned)x0 < (unsigned)d.Wi, xin1 = (unsigned)(x0 + 1) < (unsigned)d.Wi;
      const bool yin0 = (unsigned)y0 < (unsigned)d.Hi, yin1 = (unsigned)(y0 + 1) < (unsigned)d.Hi;
      const int xc0 = min(max(x0, 0), d.Wi - 1), xc1 = min(max(x0 + 1, 0), d.Wi - 1);
      const int yc0 = min(max(y0, 0), d.Hi - 1), yc1 = min(max(y0 + 1, 0), d.Hi - 1);
      const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
      const bool m00 = xin0 && yin0, m01 = xin1 && yin0, m10 = xin0 && yin1, m11 = xin1 && yin1;
      const T* p00 = srcb + ((long long)yc0 * d.Wi + xc0) * CIN;
      const T* p01 = srcb + ((long long)yc0 * d.Wi + xc1) * CIN;
      const T* p10 = srcb + ((long long)yc1 * d.Wi + xc0) * CIN;
      const T* p11 = srcb + ((long long)yc1 * d.Wi + xc1) * CIN;
      const T* pr = refb + ((long long)gh * d.Wi + gw) * CIN;
#pragma unroll
      for (int c0 = 0; c0 < CPV; c0 += CG) {
        uint4 r[CG], a[CG], b[CG], c[CG], e[CG];
#pragma unroll
        for (int k = 0; k < CG; ++k) {
          r[k] = *reinterpret_cast<const uint4*>(pr + (c0 + k) * E);
          a[k] = *reinterpret_cast<const uint4*>(p00 + (c0 + k) * E);
          b[k] = *reinterpret_cast<const uint4*>(p01 + (c0 + k) * E);
          c[k] = *reinterpret_cast<const uint4*>(p10 + (c0 + k) * E);
          e[k] = *reinterpret_cast<const uint4*>(p11 + (c0 + k) * E);
        }
#pragma unroll
        for (int k = 0; k < CG; ++k) {
          const uint4 z = make_uint4(0u, 0u, 0u, 0u);
          float fr[E], fa[E], fb[E], fc[E], fe[E], o[E];
          unpack_chunk(r[k], fr, T());
          unpack_chunk(m00 ? a[k] : z, fa, T());       // zeros padding: an outside corner contributes exactly 0
          unpack_chunk(m01 ? b[k] : z, fb, T());
          unpack_chunk(m10 ? c[k] : z, fc, T());
          unpack_chunk(m11 ? e[k] : z, fe, T());
#pragma unroll
          for (int q = 0; q < E; ++q) {
            const float wsum = ((fa[q] * w00 + fb[q] * w01) + fc[q] * w10) + fe[q] * w11;
            o[q] = fin ? fr[q] + wsum : __builtin_nanf("");
          }
          *reinterpret_cast<uint4*>(dst + (c0 + k) * 16) = pack_chunk(o, T());
        }
      }
    }
  }
  __syncthreads();

  // ---- per-lane fragment bases: voxel (lane&15) of fragment f at tap (0,0,0) ----
  int base[NF];
  int qd_[NF], qh_[NF], qw_[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int vt = ((wave / WC) * NF + f) * 16 + lr;   // voxel index inside the tile, w fastest
    const int w_ = vt % TW, h_ = (vt / TW) % TH, d_ = vt / (TW * TH);
    base[f] = ((d_ * STRIDE * HH + h_ * STRIDE) * HW + w_ * STRIDE) * VS + (TPS == 1 ? lg * 16 : 0);
    qd_[f] = q0d + d_; qh_[f] = q0h + h_; qw_[f] = q0w + w_;
  }

  const int co0 = (wave % WC) * FM * 16;              // this wave's slice of the output channels
  const unsigned char* __restrict__ wg = reinterpret_cast<const unsigned char*>(d.wgt) + (long long)co0 * 64 + lr * 64 + lg * 16;
  T* __restrict__ out = reinterpret_cast<T*>(d.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(d.res);

  float bv[FM][4];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[a][e] = (co0 + a * 16 + lg * 4 < d.Cout) ? d.bias[co0 + a * 16 + lg * 4 + e] : 0.f;

  auto run_pass = [&](auto pc) {
    constexpr int PASS = decltype(pc)::value;
    constexpr int KH = Cfg::kh_of(PASS), KW = Cfg::kw_of(PASS);
    constexpr int NT = Cfg::taps_of(PASS), NS = Cfg::steps_of(PASS), S0 = Cfg::step0_of(PASS);
    f32x4 acc[FM][NF];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // halo byte offset of step s's tap for this lane (compile-time when one tap spans the whole step); pad: the tap does not exist
    auto step_off = [&](int s, int& so, bool& pad) {
      pad = false;
      if constexpr (TPS == 1) {
        const int tap = s / SPT;
        so = (((tap / (KH * KW)) * HH + (tap / KW) % KH) * HW + tap % KW) * VS + (s % SPT) * 64;
      } else {
        const int tap = s * TPS + lg / GPT;
        pad = tap >= NT;
        so = (((tap / (KH * KW)) * HH + (tap / KW) % KH) * HW + tap % KW) * VS + (lg % GPT) * 16;
      }
    };
    if constexpr (std::is_same<T, bx3_t>::value) {
      // split pairs: two consecutive steps (2 x 4 k values per lane, hi and lo) make the operands of the full-rate 16x16x32
      // instruction; an odd last step runs on the K=16 form
      {
        // same schedule as the 16-bit path below, in units of step PAIRS: the operands of pair p + 1 are read from LDS before the
        // MFMAs of pair p (two register sets), the packed weights run two pairs ahead through a register ring
        constexpr int NP = (NS + 1) / 2;
        uint4 bq0[2][NF], bq1[2][NF], aq0[2][FM], aq1[2][FM];
        auto read_pair = [&](auto PI, uint4 (&d0)[NF], uint4 (&d1)[NF]) {
          constexpr int s = 2 * decltype(PI)::value;
          constexpr bool two = s + 1 < NS;
          int so0, so1 = 0;
          bool pad0, pad1 = false;
          step_off(s, so0, pad0);
          if constexpr (two) step_off(s + 1, so1, pad1);
          constexpr bool mp0 = (TPS > 1) && ((s + 1) * TPS > NT), mp1 = (TPS > 1) && ((s + 2) * TPS > NT);
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            d0[f] = *reinterpret_cast<const uint4*>(halo + ((mp0 && pad0) ? ZERO_OFF : base[f] + so0));
            if constexpr (two) d1[f] = *reinterpret_cast<const uint4*>(halo + ((mp1 && pad1) ? ZERO_OFF : base[f] + so1));
          }
        };
        auto load_pair = [&](auto PI, uint4 (&d0)[FM], uint4 (&d1)[FM]) {
          constexpr int s = 2 * decltype(PI)::value;
#pragma unroll
          for (int a = 0; a < FM; ++a) {
            d0[a] = *reinterpret_cast<const uint4*>(wg + ((long long)(S0 + s) * COUTP + a * 16) * 64);
            if constexpr (s + 1 < NS) d1[a] = *reinterpret_cast<const uint4*>(wg + ((long long)(S0 + s + 1) * COUTP + a * 16) * 64);
          }
        };
        load_pair(IC<0>{}, aq0[0], aq1[0]);
        if constexpr (NP > 1) load_pair(IC<1>{}, aq0[1], aq1[1]);
        read_pair(IC<0>{}, bq0[0], bq1[0]);
        static_for<NP>([&](auto PI) {
          constexpr int pi = decltype(PI)::value;
          constexpr bool two = 2 * pi + 1 < NS;
          uint4 a0[FM], a1[FM];
#pragma unroll
          for (int a = 0; a < FM; ++a) {
            a0[a] = aq0[pi & 1][a];
            if constexpr (two) a1[a] = aq1[pi & 1][a];
          }
          if constexpr (pi + 2 < NP) load_pair(IC<pi + 2>{}, aq0[pi & 1], aq1[pi & 1]);
          if constexpr (pi + 1 < NP) read_pair(IC<pi + 1>{}, bq0[(pi + 1) & 1], bq1[(pi + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (two) {
            uint4 ah[FM], al[FM];
#pragma unroll
            for (int a = 0; a < FM; ++a) bx3_pair(a0[a], a1[a], ah[a], al[a]);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              uint4 bh, bl;
              bx3_pair(bq0[pi & 1][f], bq1[pi & 1][f], bh, bl);
#pragma unroll
              for (int a = 0; a < FM; ++a) {
                acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[a]), __builtin_bit_cast(bf16x8, bh), acc[a][f], 0, 0, 0);
                acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[a]), __builtin_bit_cast(bf16x8, bl), acc[a][f], 0, 0, 0);
                acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[a]), __builtin_bit_cast(bf16x8, bh), acc[a][f], 0, 0, 0);
              }
            }
          } else {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
              for (int a = 0; a < FM; ++a) Mma3<T>::run(a0[a], bq0[pi & 1][f], acc[a][f]);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    } else {
    {
      // B operands of step s + 1 are read from LDS before the MFMAs of step s (two register sets).  Left alone hipcc emitted
      // ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma for every single MFMA, re-using one register quad: the LDS round trip (~120
      // cycles) in front of each 16-cycle instruction — the tap loop ran at a quarter of the matrix rate.
      uint4 bq[2][NF];
      auto read_b = [&](int s, uint4 (&dst)[NF]) {
        int so;
        bool pad;
        step_off(s, so, pad);
        // a padded tap must read the zero voxel itself: its weights are 0, but 0 * stale NaN bytes would poison acc
        const bool maybe_pad = (TPS > 1) && ((s + 1) * TPS > NT);
#pragma unroll
        for (int f = 0; f < NF; ++f) dst[f] = *reinterpret_cast<const uint4*>(halo + ((maybe_pad && pad) ? ZERO_OFF : base[f] + so));
      };
      // A operands (packed weights, L2-resident): a ring PDA steps ahead (the scheduling barriers below would otherwise pin
      // each load in the step that consumes it)
      constexpr int PDA = NS < 4 ? NS : 4;
      uint4 aq[PDA][FM];
      auto load_a = [&](int s, uint4 (&dst)[FM]) {
#pragma unroll
        for (int a = 0; a < FM; ++a) dst[a] = *reinterpret_cast<const uint4*>(wg + ((long long)(S0 + s) * COUTP + a * 16) * 64);
      };
#pragma unroll
      for (int p = 0; p < PDA; ++p) load_a(p, aq[p]);
      read_b(0, bq[0]);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        uint4 af[FM];
#pragma unroll
        for (int a = 0; a < FM; ++a) af[a] = aq[s % PDA][a];
        if (s + PDA < NS) load_a(s + PDA, aq[s % PDA]);
        if (s + 1 < NS) read_b(s + 1, bq[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int a = 0; a < FM; ++a) Mma3<T>::run(af[a], bq[s & 1][f], acc[a][f]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
    // ---- epilogue: bias (folded BN), ReLU, post-activation skip add, 4-channel vector store ----
    // All skip reads of the pass are requested before its first store, and the bias sits in registers (bv, loaded once per
    // kernel): on gfx9 a load issued after a store waits, through the shared in-order vmcnt, for that store's round trip.
    constexpr int pd = TR ? (PASS >> 2) & 1 : 0, ph = TR ? (PASS >> 1) & 1 : 0, pw = TR ? PASS & 1 : 0;
    constexpr int OS = TR ? 2 : 1;
    long long opix[NF];
    bool fok[NF];
    float rv[NF][FM][4];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      fok[f] = !(qd_[f] >= d.Dq || qh_[f] >= d.Hq || qw_[f] >= d.Wq);
      // class-major output (transposed convs whose consumer gathers sparsely): the 8 sub-pixel classes become 8 dense
      // volumes, so a fragment row is one full 128-byte line instead of eight 16-byte pieces of eight lines
      const long long rpix = (((long long)n * d.Do + (qd_[f] * OS + pd)) * d.Ho + (qh_[f] * OS + ph)) * d.Wo + (qw_[f] * OS + pw);
      opix[f] = (TR && d.out_classmajor) ? ((((long long)PASS * d.N + n) * d.Dq + qd_[f]) * d.Hq + qh_[f]) * d.Wq + qw_[f] : rpix;
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        const int ch = co0 + a * 16 + lg * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) rv[f][a][e] = 0.f;
        if (res && fok[f] && ch < d.Cout) load4(res + rpix * d.Cout + ch, rv[f][a]);
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (!fok[f]) continue;
#pragma unroll
      for (int a = 0; a < FM; ++a) {
        const int ch = co0 + a * 16 + lg * 4;
        if (ch >= d.Cout) continue;
        float v[4] = {acc[a][f][0], acc[a][f][1], acc[a][f][2], acc[a][f][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += bv[a][e];
          if (d.relu) v[e] = v[e] < 0.f ? 0.f : v[e];      // NaN propagates, like torch.relu
          v[e] += rv[f][a][e];
        }
        store4(out + opix[f] * d.Cout + ch, v);
      }
    }
  };
  run_pass(IC<0>{});
  if constexpr (TR) {
    run_pass(IC<1>{}); run_pass(IC<2>{}); run_pass(IC<3>{});
    run_pass(IC<4>{}); run_pass(IC<5>{}); run_pass(IC<6>{}); run_pass(IC<7>{});
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
  const int E = dtype_chunk(dtype);
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

template <typename T, int CIN, int COUTP, int TD, int TH, int TW, int STRIDE, bool TR, bool WARP, int WC = 1>
static int launch_c3(Conv3dTileDesc d, hipStream_t s) {
  using Cfg = C3Cfg<T, CIN, COUTP, TD, TH, TW, STRIDE, TR, WC>;
  constexpr size_t LDS = Cfg::LDS_BYTES;
  static_assert(LDS <= 160 * 1024, "tile does not fit LDS");
  d.ntd = (d.Dq + TD - 1) / TD; d.nth = (d.Hq + TH - 1) / TH; d.ntw = (d.Wq + TW - 1) / TW;
  const long long nblk = (long long)d.N * d.ntd * d.nth * d.ntw;
  RGBM_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv3d grid out of range");
  auto kern = conv3d_tile_kernel<T, CIN, COUTP, TD, TH, TW, STRIDE, TR, WARP, WC>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)LDS)) return rc;
  prof_begin_launch(s, d.prof_variant, d.algo_flops, d.algo_bytes);
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), LDS, s, d);
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int conv3d_tile_dims(int layer, int dtype, int* TD, int* TH, int* TW) {
  const bool b16 = dtype == BF16 || dtype == F16;
#define C3_CASE(L, CIN, COUTP, TDB, THB, TWB, TDF, THF, TWF, STRIDE, TR, WARP, WCB, WCF) \
  case L: *TD = b16 ? TDB : TDF; *TH = b16 ? THB : THF; *TW = b16 ? TWB : TWF; return 0;
  switch (layer) {
#include "conv3d_tile_table.h"
    default: break;
  }
#undef C3_CASE
  return -1;
}

// layer ids: 0..6 = conv0..conv6, 7..9 = conv7/9/11 (transposed), 10 = conv0 with fused warp
int launch_conv3d_tile(int layer, int dtype, const Conv3dTileDesc& d, hipStream_t s) {
#define C3_CASE(L, CIN, COUTP, TDB, THB, TWB, TDF, THF, TWF, STRIDE, TR, WARP, WCB, WCF)                           \
  case L:                                                                                                          \
    return dtype == BF16  ? launch_c3<unsigned short, CIN, COUTP, TDB, THB, TWB, STRIDE, TR, WARP, WCB>(d, s)      \
           : dtype == F16 ? launch_c3<f16_t, CIN, COUTP, TDB, THB, TWB, STRIDE, TR, WARP, WCB>(d, s)               \
           : dtype == BF16X3 ? launch_c3<bx3_t, CIN, COUTP, TDF, THF, TWF, STRIDE, TR, WARP, WCF>(d, s)            \
                          : launch_c3<float, CIN, COUTP, TDF, THF, TWF, STRIDE, TR, WARP, 1>(d, s);
  switch (layer) {
#include "conv3d_tile_table.h"
    default: break;
  }
#undef C3_CASE
  set_error("conv3d_tile: unknown layer id");
  return -1;
}

}  // namespace rgbm
