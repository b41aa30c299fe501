// conv0 of the cost-regularisation net (Conv3d 32->8, k3 p1 + BN + ReLU, network_v5.py:260-291) with the plane sweep
// (homo_warping + "ref + warped" fusion, network_v5.py:378-430) built on the fly — depth-sweeping, role-specialised
// bf16 kernel for gfx950.  Replaces the halo-tile kernel (conv3d_tile.hip, layer 10) whose staging, MFMA and store
// phases ran back to back: all blocks of a launch march through identical phases in lock step, so nothing overlaps.
//
// One workgroup owns a 16x16 (H x W) column of one view and sweeps it through all D depth planes:
//   * producer waves (6): one thread per voxel of the 18x18 input plane incl. halo.  The thread keeps the reference
//     feature of its pixel in registers for the whole sweep; per plane it projects the pixel with that plane's depth,
//     gathers the 4 bilinear corners of the partner view's feature map (corners outside the image are pointed at a zero
//     page: grid_sample padding_mode="zeros"), blends in fp32, rounds to bf16 and writes the 64-byte voxel into one of
//     two LDS plane slots.  The corner loads of plane z+1 are issued before plane z is blended (two half-voxel register
//     sets), so gather latency is covered by VALU work.  Halo redundancy is 324/256 = 1.27x (the 4x8x8 tile: 2.34x).
//   * consumer waves (4): MFMA 16x16x32 bf16, input-plane stationary.  Cout = 8 fills only half of the 16 MFMA rows,
//     so two depth taps share one instruction: A01[t] = rows 0-7 W(kd=0,t), rows 8-15 W(kd=1,t); A2[t] = rows 8-15
//     W(kd=2,t).  For input plane p and in-plane tap t:  X[p] += A01[t]*B,  X[p-1] += A2[t]*B  (same B register), hence
//     out[o] = rows 8-15 of X[o] (kd=1 from plane o, kd=2 from plane o+1)  +  rows 0-7 of X[o-1] (kd=0 from plane o-1).
//     18 MFMAs per 16-voxel fragment and plane instead of 27, one LDS read per two MFMAs, weights live in registers.
//     The two row halves sit in lanes 0-31 / 32-63 of the accumulator: v_permlane32_swap pairs two fragments so the
//     final add, bias, ReLU and the 8-byte stores run on all 64 lanes.
// One s_barrier per plane hands slot z&1 from the producers to the consumers; producers fill the other slot meanwhile.
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace rgbm {

namespace {

constexpr int SW_TH = 16, SW_TW = 16;
constexpr int SW_HH = SW_TH + 2, SW_HW = SW_TW + 2;
constexpr int SW_NV = SW_HH * SW_HW;             // 324 voxels per input plane
constexpr int SW_VS = 80;                        // LDS bytes per voxel: 64 data + 16 pad (conflict-free b128 rows)
constexpr int SW_SLOT = SW_NV * SW_VS;           // 25920
constexpr int SW_NPW = 6, SW_NCW = 4;            // producer / consumer waves
constexpr int SW_THREADS = (SW_NPW + SW_NCW) * 64;
constexpr int SW_LDS = 2 * SW_SLOT;

__device__ uint4 g_sweep_zero[4];                // 64 zero bytes: the "feature" of every out-of-image corner

struct SweepDesc {
  const unsigned short* feat;     // [V][H][W][32] bf16
  const unsigned short* wgt;      // [18][16][4][8] bf16 (conv0_sweep_pack)
  const float* bias;              // [16] folded BN shift
  const float* homog;             // [V][12]
  const float* depths;            // [B][D]
  unsigned short* out;            // [N][D][H][W][8] bf16
  int N, D, H, W, v0, V, B, nth, ntw, relu, dbg;
};

__device__ __forceinline__ void sweep_ixy(const float* __restrict__ hm, float x, float y, float depth, int H, int W, float& ix,
                                          float& iy) {
  // same arithmetic as warp_ixy (conv3d_tile.hip) / build_volume: homography, perspective divide, the reference's
  // align_corners=True normalisation followed by grid_sample's align_corners=False un-normalisation
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

struct Corner {                     // everything the blend of one plane needs besides the gathered data
  const unsigned short* p[4];       // 4 corner pointers (zero page when outside)
  float w[4];                       // bilinear weights (NaN when the projection is not finite, like the reference)
};

__device__ __forceinline__ uint4 blend_chunk(const uint4& r, const uint4& a, const uint4& b, const uint4& c, const uint4& e,
                                             const float* w) {
  float fr[8], fa[8], fb[8], fc[8], fe[8], o[8];
  unpack_chunk(r, fr, (unsigned short)0);
  unpack_chunk(a, fa, (unsigned short)0);
  unpack_chunk(b, fb, (unsigned short)0);
  unpack_chunk(c, fc, (unsigned short)0);
  unpack_chunk(e, fe, (unsigned short)0);
#pragma unroll
  for (int q = 0; q < 8; ++q) o[q] = fr[q] + (((fa[q] * w[0] + fb[q] * w[1]) + fc[q] * w[2]) + fe[q] * w[3]);
  return pack_chunk(o, (unsigned short)0);
}

__device__ __forceinline__ f32x4 mma_bf16(const uint4& a, const uint4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace

__global__ __launch_bounds__(SW_THREADS) void conv0_sweep_kernel(const SweepDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char planes[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // XCD-aware tile order: every XCD walks a contiguous run of tiles (whole views) so the partner feature maps its CUs
  // gather from stay in that XCD's L2
  const int nblk = gridDim.x, bq = nblk >> 3, br = nblk & 7, xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  int t = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
  const int tw = t % d.ntw; t /= d.ntw;
  const int th = t % d.nth; t /= d.nth;
  const int n = t;
  const int h0 = th * SW_TH, w0 = tw * SW_TW;
  const int D = d.D, H = d.H, W = d.W;
  const int vv = d.v0 + n;

  if (wave < SW_NPW) {
    // ------------------------------------------------------------------ producers
    const int pv = tid;                                  // voxel of the 18x18 plane
    const bool act = pv < SW_NV;
    const int hh = pv / SW_HW, hw = pv - hh * SW_HW;
    const int gh = h0 - 1 + hh, gw = w0 - 1 + hw;
    const bool inb = act && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
    const int partner = (vv + d.B) % d.V, bb = vv % d.B;
    const float* __restrict__ hm = d.homog + (long long)vv * 12;
    const float* __restrict__ dep = d.depths + (long long)bb * D;
    const unsigned short* __restrict__ srcb = d.feat + (long long)partner * H * W * 32;
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_sweep_zero);
    unsigned char* dst0 = planes + pv * SW_VS;

    uint4 ref[4];
    {
      const unsigned short* pr = inb ? d.feat + (((long long)vv * H + gh) * W + gw) * 32 : zero;
#pragma unroll
      for (int k = 0; k < 4; ++k) ref[k] = *reinterpret_cast<const uint4*>(pr + k * 8);
    }

    auto corners = [&](int z, Corner& c) {
      float ix, iy;
      sweep_ixy(hm, (float)gw, (float)gh, dep[z], H, W, ix, iy);
      const bool fin = isfinite(ix) && isfinite(iy);
      ix = fin ? fminf(fmaxf(ix, -4.f), 1.0e6f) : 0.f;
      iy = fin ? fminf(fmaxf(iy, -4.f), 1.0e6f) : 0.f;
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float tx = ix - fx, ty = iy - fy;
      const bool xin0 = (unsigned)x0 < (unsigned)W, xin1 = (unsigned)(x0 + 1) < (unsigned)W;
      const bool yin0 = (unsigned)y0 < (unsigned)H, yin1 = (unsigned)(y0 + 1) < (unsigned)H;
      const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
      const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
      const float nanv = __builtin_nanf("");
      // a voxel outside the image is conv zero padding: exact 0 whatever the projection says
      const bool usew = fin || !inb;
      c.w[0] = usew ? (1.f - tx) * (1.f - ty) : nanv;
      c.w[1] = usew ? tx * (1.f - ty) : nanv;
      c.w[2] = usew ? (1.f - tx) * ty : nanv;
      c.w[3] = usew ? tx * ty : nanv;
      c.p[0] = (inb && xin0 && yin0) ? srcb + ((long long)yc0 * W + xc0) * 32 : zero;
      c.p[1] = (inb && xin1 && yin0) ? srcb + ((long long)yc0 * W + xc1) * 32 : zero;
      c.p[2] = (inb && xin0 && yin1) ? srcb + ((long long)yc1 * W + xc0) * 32 : zero;
      c.p[3] = (inb && xin1 && yin1) ? srcb + ((long long)yc1 * W + xc1) * 32 : zero;
    };

    Corner cur, nxt;
    uint4 ga[2][4], gb[2][4];                            // [chunk][corner] for chunks 0-1 (ga) and 2-3 (gb)
    if (act) {
      corners(0, cur);
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) ga[k][q] = *reinterpret_cast<const uint4*>(cur.p[q] + k * 8);
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) gb[k][q] = *reinterpret_cast<const uint4*>(cur.p[q] + (2 + k) * 8);
    }
    for (int z = 0; z <= D; ++z) {
      if (act && z < D) {
        unsigned char* dst = dst0 + (z & 1) * SW_SLOT;
        const bool more = z + 1 < D;
        if (more) corners(z + 1, nxt);
        if (!(d.dbg & 1)) {
#pragma unroll
          for (int k = 0; k < 2; ++k)
            *reinterpret_cast<uint4*>(dst + k * 16) = blend_chunk(ref[k], ga[k][0], ga[k][1], ga[k][2], ga[k][3], cur.w);
        }
        if (more) {
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) ga[k][q] = *reinterpret_cast<const uint4*>(nxt.p[q] + k * 8);
        }
        if (!(d.dbg & 1)) {
#pragma unroll
          for (int k = 0; k < 2; ++k)
            *reinterpret_cast<uint4*>(dst + (2 + k) * 16) = blend_chunk(ref[2 + k], gb[k][0], gb[k][1], gb[k][2], gb[k][3], cur.w);
        }
        if (more) {
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) gb[k][q] = *reinterpret_cast<const uint4*>(nxt.p[q] + (2 + k) * 8);
          cur = nxt;
        }
      }
      // the plane just written must be visible before the consumers are released; prefetched gathers stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    // ------------------------------------------------------------------ consumers
    const int cw = wave - SW_NPW;
    const int lr = lane & 15, lg = lane >> 4;
    uint4 A01[9], A2[9];
    {
      const uint4* wq = reinterpret_cast<const uint4*>(d.wgt);
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        A01[s] = wq[(s * 16 + lr) * 4 + lg];
        A2[s] = wq[((9 + s) * 16 + lr) * 4 + lg];
      }
    }
    // after the lane-half swap a lane holds: fragment (lg < 2 ? first : second of the pair), voxel lr, channels (lg&1)*4..+3
    const int ch = (lg & 1) * 4;
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = d.bias[ch + r];
    const int ow = w0 + lr;
    int oh[2];
    bool ook[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      oh[pr] = h0 + cw * 4 + pr * 2 + (lg >> 1);
      ook[pr] = oh[pr] < H && ow < W;
    }
    const int boff = ((cw * 4) * SW_HW + lr) * SW_VS + lg * 16;     // fragment 0, tap (0,0)

    f32x4 Xp[4], Lp[2];
#pragma unroll
    for (int f = 0; f < 4; ++f) Xp[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    Lp[0] = Lp[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto emit = [&](int o) {           // out plane o from Xp (= X[o]) and Lp (= rows 0-7 of X[o-1]); leaves Lp = rows 0-7 of X[o]
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        f32x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Xp[2 * pr][r]), __float_as_uint(Xp[2 * pr + 1][r]), false, false);
          lo[r] = __uint_as_float(sw[0]);      // lanes 0-31: rows 0-7 of frag 2pr, lanes 32-63: rows 0-7 of frag 2pr+1
          hi[r] = __uint_as_float(sw[1]);      // rows 8-15 likewise
        }
        if (o >= 0 && ook[pr]) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = (hi[r] + Lp[pr][r]) + bias[r];
            if (d.relu) v[r] = v[r] < 0.f ? 0.f : v[r];            // NaN propagates, like torch.relu
          }
          store4(d.out + ((((long long)n * D + o) * H + oh[pr]) * W + ow) * 8 + ch, v);
        }
        Lp[pr] = lo;
      }
    };

    for (int z = 0; z <= D; ++z) {
      if (z >= 1) {
        const int p = z - 1;
        const unsigned char* slot = planes + (p & 1) * SW_SLOT + boff;
        f32x4 Xn[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) Xn[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(d.dbg & 2)) {
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              const uint4 b = *reinterpret_cast<const uint4*>(slot + ((f + tp / 3) * SW_HW + tp % 3) * SW_VS);
              Xn[f] = mma_bf16(A01[tp], b, Xn[f]);
              Xp[f] = mma_bf16(A2[tp], b, Xp[f]);
            }
          }
        }
        emit(p - 1);                   // X[p-1] is complete once plane p has contributed its kd=2 taps
#pragma unroll
        for (int f = 0; f < 4; ++f) Xp[f] = Xn[f];
      }
      __builtin_amdgcn_s_barrier();
    }
    emit(D - 1);                       // plane D is zero padding: X[D-1] is already complete
  }
}

// Pack conv0 weights [8][32][27] (x folded BN scale) into the consumer's A-fragment order [18][16 rows][4 k-groups][8]:
// steps 0..8 = A01 of in-plane tap t (rows 0-7 kd=0, rows 8-15 kd=1), steps 9..17 = A2 (rows 8-15 kd=2, rows 0-7 zero).
void conv0_sweep_pack(const float* w, const float* scale, std::vector<float>& packed) {
  packed.assign((size_t)18 * 16 * 4 * 8, 0.f);
  for (int s = 0; s < 18; ++s)
    for (int row = 0; row < 16; ++row) {
      const int tpl = s % 9, o = row & 7;
      int kd;
      if (s < 9) kd = row < 8 ? 0 : 1;
      else { if (row < 8) continue; kd = 2; }
      const int widx = kd * 9 + tpl;                   // tpl = kh*3 + kw
      for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 8; ++e) {
          const int c = g * 8 + e;
          packed[(((size_t)s * 16 + row) * 4 + g) * 8 + e] = w[((long long)o * 32 + c) * 27 + widx] * (scale ? scale[o] : 1.f);
        }
    }
}

int launch_conv0_sweep(const Conv3dTileDesc& t, hipStream_t s) {
  SweepDesc d;
  d.feat = reinterpret_cast<const unsigned short*>(t.feat);
  d.wgt = reinterpret_cast<const unsigned short*>(t.wgt);
  d.bias = t.bias; d.homog = t.homog; d.depths = t.depths;
  d.out = reinterpret_cast<unsigned short*>(t.out);
  d.N = t.N; d.D = t.Di; d.H = t.Hi; d.W = t.Wi; d.v0 = t.v0; d.V = t.V; d.B = t.B; d.relu = t.relu;
  d.nth = (d.H + SW_TH - 1) / SW_TH; d.ntw = (d.W + SW_TW - 1) / SW_TW;
  d.dbg = g_debug_flags;
  RGBM_REQUIRE(d.feat && d.wgt && d.bias && d.homog && d.depths && d.out && d.D >= 1 && t.Cout == 8, "conv0 sweep arguments");
  const long long nblk = (long long)d.N * d.nth * d.ntw;
  RGBM_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv0 sweep grid out of range");
  static bool attr_done = false;
  if (!attr_done) {
    RGBM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv0_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS));
    attr_done = true;
  }
  prof_begin_launch(s, t.prof_variant, t.algo_flops, t.algo_bytes);
  hipLaunchKernelGGL(conv0_sweep_kernel, dim3((unsigned)nblk), dim3(SW_THREADS), SW_LDS, s, d);
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
