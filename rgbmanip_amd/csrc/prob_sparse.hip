// Sparse tail of the cost-regularisation net (bf16 nets): conv11 (ConvTranspose3d 16->8, k3 s2 p1 op1 + BN + ReLU), the
// skip add u11 = c0 + conv11(u9), the prob conv (Conv3d 8->1, k3 p1), softmax over depth and the depth regression
// (network_v5.py:278-299, 449-455) — evaluated ONLY where the network consumes them.  prob is gathered at the P = 1024
// chosen pixels of a view, so of the 24 x 224 x 224 u11 volume only the 3 x 3 pixel neighbourhoods of those pixels are
// ever read: at most 9216 of 50176 pixel columns (18 %), far fewer for a real (connected) object mask.  The dense
// pipeline it replaces spent 9.1 ms per batch writing u11 (conv11 is bound by partial-line memory instructions, 2.65 TB/s)
// plus 2.1 ms gathering from it.
//
// One wave per chosen pixel.  The transposed conv is computed per output-parity class exactly like the halo-tile kernel
// (conv3d_tile.hip, TR mode: 8 classes with 1..8 taps, two taps of 16 channels per 16x16x32 MFMA step, the same packed
// weight fragments), but its B operand is gathered straight from u9 instead of a staged halo: lane (voxel, k-group) loads
// the 16-byte half voxel its MFMA lane needs.  A class has 12, 24 or 48 neighbourhood voxels = 1..3 fragments.  u11 itself
// is never stored: the prob conv has ONE output channel, so a neighbourhood voxel (z, kh, kw) contributes to at most three
// logits (z - 1, z, z + 1) through the channel dot products q_kd = sum_c wprob[kd][kh][kw][c] * u11[c]; the lanes that hold
// the voxel's fp32 channels form those three numbers (12 FMAs + one cross-lane add) and only they go to LDS (3 x 9 x D
// floats per wave instead of the 6.9 KB neighbourhood).  Then lane z adds its 27 partial sums in a fixed order, and the wave
// reduces softmax and depth.
//
// What bounds it (round 3): the instruction stream.  About 2500 VALU + 1000 SALU instructions per point (index arithmetic and
// bounds of every gather, the epilogue of every fragment), 512 points per SIMD: 2.9 ms whether 5 or 8 waves per SIMD are
// resident and whichever XCD a point lands on.  A variant with the point's coordinates forced into SGPRs (readfirstlane),
// buffer loads with out-of-range offsets instead of selects and branch-free index arithmetic measured 3.2 ms — not kept.
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace rgbm {

extern int g_debug_flags;

namespace {

constexpr int PS_DMAX = 24;

struct ProbSparseDesc {
  const void* u9;               // [Vc][D/2][H/2][W/2][16] in the storage type
  const void* c0;               // [Vc][D][H][W][8]
  const void* w11;              // conv3d_tile_pack(conv11) in the 16-bit step geometry: [14 steps][16][4][8] (bf16x3: hi operands, then lo)
  const float* bias11;          // [16] folded BN shift
  const float* wprob;           // [27][8]
  const int* choose;            // [V][P]
  const float* depths;          // [B][D]
  float* prob;                  // [V][P][D]
  float* depth_out;             // [V][P]
  int v0, Vc, B, P, D, H, W;
};

template <int N> struct PC { static constexpr int value = N; };

template <typename T> __device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, const f32x4& c);
template <> __device__ __forceinline__ f32x4 mma16<unsigned short>(const uint4& a, const uint4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<f16_t>(const uint4& a, const uint4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
constexpr int PS_W11_OPS = 14 * 16 * 4;      // 16-byte weight operands of conv11 (bf16x3: the lo operands follow the hi ones)

}  // namespace

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? 6 : 5, 8))) void prob_sparse_kernel(const ProbSparseDesc d) {
  __shared__ __attribute__((aligned(16))) float Q[4][PS_DMAX * 9 * 4];      // q_kd (kd = 0..2, one pad float) of the 3x3xD neighbourhood, per wave
  __shared__ __attribute__((aligned(16))) float wp[27 * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 27 * 8; i += 256) wp[i] = d.wprob[i];
  const int D = d.D, H = d.H, W = d.W, Dq = D >> 1, Hq = H >> 1, Wq = W >> 1;
  // XCD-aware block order (round 5): hardware block b runs on XCD b % 8, and the 256 blocks of a view share its c0 / u9 lines.  In
  // launch order every XCD touched every view (c0 crossed the fabric into up to eight L2s: 11.9 GB of fetches per step for a tensor
  // whose needed part is a fraction of its 9.9 GB); XCD x now owns a contiguous run of blocks, i.e. whole views.
  const long long nb = gridDim.x, bq = nb >> 3, br = nb & 7;
  const long long xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  const long long vblk = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
  const long long pidx = vblk * 4 + wave;                                // point index within the chunk (wave-uniform)
  const bool active = pidx < (long long)d.Vc * d.P;
  const int vl = active ? (int)(pidx / d.P) : 0;
  const int v = d.v0 + vl;
  const long long o = (long long)v * d.P + (pidx - (long long)vl * d.P);
  const int pix = active ? d.choose[o] : 0;
  const int y = pix / W, x = pix - y * W;
  float* Qw = Q[wave];
  for (int i = lane * 4; i < D * 36; i += 256)                          // neighbours outside the image / volume: zero padding
    *reinterpret_cast<f32x4*>(Qw + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  constexpr bool X3 = std::is_same<T, bx3_t>::value;      // split pairs: 8 channels = two 16-byte chunks {hi, lo}, three products
  const uint4* wq = reinterpret_cast<const uint4*>(d.w11);
  const T* u9v = reinterpret_cast<const T*>(d.u9) + (long long)vl * Dq * Hq * Wq * 16;
  const T* c0v = reinterpret_cast<const T*>(d.c0) + (long long)vl * D * H * W * 8;

  auto run_class = [&](auto pc) {
    constexpr int PASS = decltype(pc)::value;
    constexpr int pd = (PASS >> 2) & 1, ph = (PASS >> 1) & 1, pw = PASS & 1;
    constexpr int KH = 1 + ph, KW = 1 + pw, NT = (1 + pd) * KH * KW, NS = (NT + 1) / 2;
    constexpr int S0 = PASS == 0 ? 0 : PASS == 1 ? 1 : PASS == 2 ? 2 : PASS == 3 ? 3 : PASS == 4 ? 5 : PASS == 5 ? 6 : PASS == 6 ? 8 : 10;
    // neighbourhood voxels of this parity class: nr x nc pixel columns (1 or 2 each way) x D/2 depths, enumerated column
    // by column so that the valid slots are contiguous and empty fragments can be skipped (wave-uniform: y, x are)
    const bool ym = (y & 1) == ph, xm = (x & 1) == pw;                  // the centre row / column has this parity
    const int nr = ym ? 1 : 2, nc = xm ? 1 : 2;
    const int nvox = nr * nc * 12;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      if (f * 16 >= nvox) break;
      const int slot = f * 16 + lr;
      const int ci = slot / 12, zi = slot - ci * 12;
      const int rs = ym ? 0 : (ci & 1), cs = ym ? ci : (ci >> 1);       // ci = rs + nr*cs
      const int oz = pd + 2 * zi;
      const int yy = ym ? y : (rs ? y + 1 : y - 1);
      const int xx = xm ? x : (cs ? x + 1 : x - 1);
      const bool valid = active && slot < nvox && oz < D && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const int qd = oz >> 1, qh = yy >> 1, qw = xx >> 1;              // input (u9) voxel of tap (0,0,0)
      // every gather of the fragment is requested before the first MFMA (masked lanes read voxel 0 and are zeroed)
      uint4 bv[NS], bv2[X3 ? NS : 1];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        // MFMA step s: lane groups 0,1 carry tap 2s, groups 2,3 tap 2s+1; each group one 16-byte half of the 16 channels
        const int t0 = 2 * s, t1 = 2 * s + 1;
        const bool hi = lg >= 2;
        const int td = hi ? t1 / (KH * KW) : t0 / (KH * KW);
        const int th = hi ? (t1 / KW) % KH : (t0 / KW) % KH;
        const int tw = hi ? t1 % KW : t0 % KW;
        const bool pad = hi ? (t1 >= NT) : false;
        const int iz = qd + td, iy = qh + th, ix = qw + tw;
        const bool inb = valid && !pad && iz < Dq && iy < Hq && ix < Wq;  // beyond the input grid: the conv's zero halo
        const int off = inb ? (((iz * Hq + iy) * Wq + ix) * 16 + (lg & 1) * 8) : 0;       // per-view offsets fit 32 bits (launcher checks)
        const uint4 t = *reinterpret_cast<const uint4*>(u9v + off);
        bv[s] = inb ? t : make_uint4(0u, 0u, 0u, 0u);
        if constexpr (X3) {                                             // channels +4..+7 of the lane's 8: the second chunk
          const uint4 t2 = *reinterpret_cast<const uint4*>(u9v + off + 4);
          bv2[s] = inb ? t2 : make_uint4(0u, 0u, 0u, 0u);
        }
      }
      const bool wr = valid && lg < 2;
      const int ch = (lg & 1) * 4;
      float cv[4];
      load4(c0v + (wr ? (((oz * H + yy) * W + xx) * 8 + ch) : 0), cv);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (X3) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          uint4 bh, bl;
          bx3_pair(bv[s], bv2[s], bh, bl);
          const uint4 wh = wq[((S0 + s) * 16 + lr) * 4 + lg], wl = wq[PS_W11_OPS + ((S0 + s) * 16 + lr) * 4 + lg];
          acc = mma16<unsigned short>(wl, bh, acc);
          acc = mma16<unsigned short>(wh, bl, acc);
          acc = mma16<unsigned short>(wh, bh, acc);
        }
      } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc = mma16<T>(wq[((S0 + s) * 16 + lr) * 4 + lg], bv[s], acc);
      }
      {
        // u11 channels ch..ch+3 of the lane's voxel -> their share of the three channel dot products; the other half of the
        // channels sits 16 lanes away (lg ^ 1).  Lanes without a voxel (wr false) take part in the exchange with zeros.
        const int col = wr ? (yy - y + 1) * 3 + (xx - x + 1) : 0;
        float q[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[e] + d.bias11[ch + e];
          t = t < 0.f ? 0.f : t;                                        // NaN propagates, like torch.relu
          const float u = cv[e] + t;                                    // skip add is post-ReLU (network_v5.py:289)
#pragma unroll
          for (int kd = 0; kd < 3; ++kd) q[kd] = fmaf(u, wp[(kd * 9 + col) * 8 + ch + e], q[kd]);
        }
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) q[kd] += __shfl_xor(q[kd], 16);  // channels 0..3 + channels 4..7 (same order in both lanes)
        if (wr && lg == 0) *reinterpret_cast<f32x4*>(Qw + (oz * 9 + col) * 4) = f32x4{q[0], q[1], q[2], 0.f};
      }
    }
  };
  run_class(PC<0>{}); run_class(PC<1>{}); run_class(PC<2>{}); run_class(PC<3>{});
  run_class(PC<4>{}); run_class(PC<5>{}); run_class(PC<6>{}); run_class(PC<7>{});
  __syncthreads();

  // ---- logits at the D depths of this pixel (lane = depth): 27 partial sums in a fixed order, then softmax and depth
  // regression across the wave ----
  float logit = -INFINITY;
  if (lane < D) {
    float acc = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
      const int zz = lane + kd - 1;
      if ((unsigned)zz >= (unsigned)D) continue;
#pragma unroll
      for (int j = 0; j < 9; ++j) acc += Qw[(zz * 9 + j) * 4 + kd];
    }
    logit = acc;
  }
  float m = logit;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  // a NaN logit must poison the whole pixel like torch.softmax does (fmaxf drops NaNs)
  float nanflag = (lane < D && logit != logit) ? 1.f : 0.f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) nanflag += __shfl_xor(nanflag, off);
  float e = lane < D ? expf(logit - m) : 0.f;
  float sum = e;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  float pr = e * (1.f / sum);
  if (nanflag > 0.f) pr = __builtin_nanf("");
  float dep = lane < D ? pr * d.depths[(v % d.B) * D + lane] : 0.f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) dep += __shfl_xor(dep, off);
  if (active) {
    if (lane < D) d.prob[o * D + lane] = pr;
    if (lane == 0) d.depth_out[o] = dep;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Round 6: the point kernel restructured (the default; debug flag 536870912 = the kernel above, for A/B).
//
// The round-3..5 kernel spends ~3500 instructions per point (waiting 0.74 with eight waves per SIMD) on per-lane index arithmetic: every
// one of its sixteen MFMA fragments recomputes voxel coordinates, bounds and gather offsets per lane, two of four lane groups idle in the
// epilogue, and twelve-voxel fragments fill 12 of 16 columns.  On the benchmark's masks a view's 1024 chosen pixels are 2-3 % of a crop-filling
// mask (1-2 per 8 x 8 pixel tile), so sharing a tile's halo between its points buys nothing and would read all of c0; what is left is to spend
// fewer instructions per point:
//   * the point's u9 neighbourhood (3 x 3 input columns x D/2 depths x 16 channels) and c0 neighbourhood (3 x 3 x D x 8) are loaded ONCE
//     with lane-linear indices (4 + 4 sixteen-byte loads per lane, all in flight together) into a wave-private LDS block; everything after
//     that reads LDS at compile-time offsets;
//   * the code is specialised on the parities of (y, x) (four wave-uniform variants): which input rows / columns and which kernel taps an
//     output column needs is then a compile-time table (ps_nt / ps_rel / ps_k below);
//   * one MFMA covers BOTH depth parities of an output column: K = 32 = [u9(iz) 16 channels | u9(iz + 1) 16 channels], rows 0-7 = the
//     even output depths 2 iz (tap kd = 1 on u9(iz)), rows 8-15 = the odd ones 2 iz + 1 (kd = 2 on u9(iz), kd = 0 on u9(iz + 1)); one
//     MFMA per in-plane tap, 16-25 per point, all four lane groups carry output voxels in the epilogue;
//   * the prob conv's channel dot products stay in registers (three running sums per lane over the nine columns), 72 floats go through
//     LDS once for the depth shift.
// Same arithmetic per voxel (fp32 accumulation, fp32 u11, fp32 dot products); the summation ORDER of taps and columns differs from the
// kernel above, so the two agree to fp32 rounding, not bit for bit (tests/test_gpu_kernels.py::test_prob_sparse_kernels_agree).
namespace {

template <typename T> struct Ps2 {
  static constexpr bool X3 = std::is_same<T, bx3_t>::value;
  static constexpr int EB = X3 ? 4 : 2;                      // bytes per element
  static constexpr int VB = 16 * EB, CB = 8 * EB;            // bytes of a u9 voxel (16 channels) / a c0 voxel (8 channels)
  static constexpr int UCH = VB / 16, CCH = CB / 16;         // 16-byte chunks per voxel
  static constexpr int USLOT = PS_DMAX / 2 + 1;              // depth slots per u9 column: D/2 inputs + one zero slot behind them
  static constexpr int U_BYTES = 9 * USLOT * VB, C_BYTES = 9 * PS_DMAX * CB, S_BYTES = 3 * PS_DMAX * 4;
  static constexpr int WAVE_BYTES = U_BYTES + C_BYTES + S_BYTES;
  static constexpr int NU = 9 * (PS_DMAX / 2) * UCH, NC = 9 * PS_DMAX * CCH;      // lane-loads of the two neighbourhoods
  static constexpr int RU = (NU + 63) / 64, RC = (NC + 63) / 64;
};
constexpr int PS2_WP_BYTES = 27 * 8 * 4;
static_assert(PS2_WP_BYTES % 16 == 0 && Ps2<unsigned short>::WAVE_BYTES % 16 == 0 && Ps2<bx3_t>::WAVE_BYTES % 16 == 0, "16-byte aligned LDS blocks");

// ConvTranspose3d k3 s2 p1 op1 along one axis, for the three output coordinates c - 1 + r (r = 0..2) around a point at c = 2 h + par, with the
// staged inputs starting at hb = (c - 1) >> 1 = h - 1 + par: output 2 h + m (m = par - 1 + r) reads input h + m / 2 with kernel index 1 if m is
// even, inputs h + (m - 1) / 2 (index 2) and h + (m + 1) / 2 (index 0) if m is odd.
constexpr int ps_nt(int par, int r) { return ((par + r + 1) & 1) ? 2 : 1; }
constexpr int ps_rel(int par, int r, int j) { const int m = par - 1 + r; return ((m & 1) ? (m - 1) / 2 + j : m / 2) + 1 - par; }
constexpr int ps_k(int par, int r, int j) { const int m = par - 1 + r; return (m & 1) ? (j == 0 ? 2 : 0) : 1; }
static_assert(ps_rel(0, 0, 0) == 0 && ps_rel(0, 0, 1) == 1 && ps_rel(0, 1, 0) == 1 && ps_rel(0, 2, 0) == 1 && ps_rel(0, 2, 1) == 2, "even coordinate");
static_assert(ps_rel(1, 0, 0) == 0 && ps_rel(1, 1, 0) == 0 && ps_rel(1, 1, 1) == 1 && ps_rel(1, 2, 0) == 1, "odd coordinate");
static_assert(ps_nt(0, 0) == 2 && ps_nt(0, 1) == 1 && ps_nt(0, 2) == 2 && ps_nt(1, 0) == 1 && ps_nt(1, 1) == 2 && ps_nt(1, 2) == 1, "taps per output");

__device__ __forceinline__ float ps_xor16(float v) {      // the value of lane ^ 16 (ds_swizzle, bit mode: and 0x1f, or 0, xor 0x10)
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401f));
}

template <typename T, int PY, int PX>
__device__ __forceinline__ void ps2_columns(const unsigned char* U, const unsigned char* Cn, const float* wp, const uint4* wq, unsigned colmask,
                                            int lr, int lg, unsigned bo, unsigned co, const float (&b11)[4], float (&S)[3]) {
  using P = Ps2<T>;
  // the nine in-plane taps' A operands (split pairs: hi, then lo): rows = MFMA row lr, K group lg
  uint4 Ah[9], Al[P::X3 ? 9 : 1];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    Ah[t] = wq[(t * 16 + lr) * 4 + lg];
    if constexpr (P::X3) Al[t] = wq[9 * 16 * 4 + (t * 16 + lr) * 4 + lg];
  }
  const int ch = (lg & 1) * 4;
  auto column = [&](auto rc, auto cc) {
    constexpr int R = decltype(rc)::value, Cc = decltype(cc)::value, col = R * 3 + Cc;
    if (!((colmask >> col) & 1u)) return;                        // outside the image (wave-uniform): the prob conv's zero padding
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jr = 0; jr < ps_nt(PY, R); ++jr)
#pragma unroll
      for (int jc = 0; jc < ps_nt(PX, Cc); ++jc) {
        const int ucol = ps_rel(PY, R, jr) * 3 + ps_rel(PX, Cc, jc), t9 = ps_k(PY, R, jr) * 3 + ps_k(PX, Cc, jc);
        const unsigned char* bp = U + ucol * (P::USLOT * P::VB) + bo;
        if constexpr (P::X3) {
          uint4 bh, bl;
          bx3_pair(*reinterpret_cast<const uint4*>(bp), *reinterpret_cast<const uint4*>(bp + 16), bh, bl);
          acc = mma16<unsigned short>(Al[t9], bh, acc);
          acc = mma16<unsigned short>(Ah[t9], bl, acc);
          acc = mma16<unsigned short>(Ah[t9], bh, acc);
        } else {
          acc = mma16<T>(Ah[t9], *reinterpret_cast<const uint4*>(bp), acc);
        }
      }
    // u11 channels ch .. ch + 3 of this lane's voxel (depth 2 lr + (lg >> 1) of the column) -> their share of the three channel dot
    // products; the other four channels sit 16 lanes away
    float cv[4];
    load4(reinterpret_cast<const T*>(Cn + col * (PS_DMAX * P::CB) + co), cv);
    const float* wl = wp + col * 8 + ch;
    float q[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = acc[e] + b11[e];
      t = t < 0.f ? 0.f : t;                                     // NaN propagates, like torch.relu
      const float u = cv[e] + t;                                 // the skip add is post-ReLU (network_v5.py:289)
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) q[kd] = fmaf(u, wl[kd * 72 + e], q[kd]);
    }
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) S[kd] += q[kd] + ps_xor16(q[kd]);      // (a + b is the same number in both lanes of the pair)
  };
  column(PC<0>{}, PC<0>{}); column(PC<0>{}, PC<1>{}); column(PC<0>{}, PC<2>{});
  column(PC<1>{}, PC<0>{}); column(PC<1>{}, PC<1>{}); column(PC<1>{}, PC<2>{});
  column(PC<2>{}, PC<0>{}); column(PC<2>{}, PC<1>{}); column(PC<2>{}, PC<2>{});
}

}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void prob_sparse2_kernel(const ProbSparseDesc d) {
  using P = Ps2<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char ps_lds[];
  float* wp = reinterpret_cast<float*>(ps_lds);                  // [27][8], shared by the four waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 27 * 8; i += 256) wp[i] = d.wprob[i];
  unsigned char* U = ps_lds + PS2_WP_BYTES + wave * P::WAVE_BYTES;      // [9 columns][USLOT depths][16 channels]
  unsigned char* Cn = U + P::U_BYTES;                                    // [9 columns][PS_DMAX depths][8 channels]
  float* Sw = reinterpret_cast<float*>(Cn + P::C_BYTES);                 // [3 kd][PS_DMAX]
  const int D = d.D, H = d.H, W = d.W, Dq = D >> 1, Hq = H >> 1, Wq = W >> 1;
  // XCD-aware block order (see prob_sparse_kernel)
  const long long nb = gridDim.x, bq = nb >> 3, br = nb & 7;
  const long long xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  const long long vblk = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
  const long long pidx = vblk * 4 + wave;                                // point index within the chunk (wave-uniform)
  const bool active = pidx < (long long)d.Vc * d.P;
  const int vl = active ? (int)(pidx / d.P) : 0;
  const int v = d.v0 + vl;
  const long long o = (long long)v * d.P + (pidx - (long long)vl * d.P);
  const int pix = active ? __builtin_amdgcn_readfirstlane(d.choose[o]) : 0;
  const int y = pix / W, x = pix - y * W;
  const int hb = (y - 1) >> 1, wb = (x - 1) >> 1;                        // first staged input row / column (-1 at the border: zeros)
  const unsigned char* u9v = reinterpret_cast<const unsigned char*>(d.u9) + (long long)vl * Dq * Hq * Wq * P::VB;
  const unsigned char* c0v = reinterpret_cast<const unsigned char*>(d.c0) + (long long)vl * D * H * W * P::CB;

  // ---- the two neighbourhoods: every load of the point is requested before the first one is used ----
  uint4 ub[P::RU], cb[P::RC];
#pragma unroll
  for (int r = 0; r < P::RU; ++r) {
    const int i = r * 64 + lane;
    const int col = i / ((PS_DMAX / 2) * P::UCH), rem = i - col * ((PS_DMAX / 2) * P::UCH);
    const int iz = rem / P::UCH, chn = rem - iz * P::UCH;
    const int row = hb + col / 3, cx = wb + col % 3;
    const bool ok = active && i < P::NU && iz < Dq && (unsigned)row < (unsigned)Hq && (unsigned)cx < (unsigned)Wq;
    const unsigned off = ok ? (unsigned)(((iz * Hq + row) * Wq + cx) * P::VB + chn * 16) : 0u;      // per-view offsets fit 32 bits (launcher checks)
    const uint4 t = *reinterpret_cast<const uint4*>(u9v + off);
    ub[r] = ok ? t : make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int r = 0; r < P::RC; ++r) {
    const int i = r * 64 + lane;
    const int col = i / (PS_DMAX * P::CCH), rem = i - col * (PS_DMAX * P::CCH);
    const int oz = rem / P::CCH, chn = rem - oz * P::CCH;
    const int yy = y - 1 + col / 3, xx = x - 1 + col % 3;
    const bool ok = active && i < P::NC && oz < D && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    const unsigned off = ok ? (unsigned)(((oz * H + yy) * W + xx) * P::CB + chn * 16) : 0u;
    const uint4 t = *reinterpret_cast<const uint4*>(c0v + off);
    cb[r] = ok ? t : make_uint4(0u, 0u, 0u, 0u);
  }
  if (lane < 9 * P::UCH)                                                 // the zero slot behind the last input depth (tap kd = 0 of the last odd output)
    *reinterpret_cast<uint4*>(U + ((lane / P::UCH) * P::USLOT + Dq) * P::VB + (lane % P::UCH) * 16) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int r = 0; r < P::RU; ++r) {
    const int i = r * 64 + lane;
    const int col = i / ((PS_DMAX / 2) * P::UCH), rem = i - col * ((PS_DMAX / 2) * P::UCH);
    const int iz = rem / P::UCH, chn = rem - iz * P::UCH;
    if (i < P::NU && iz < Dq) *reinterpret_cast<uint4*>(U + (col * P::USLOT + iz) * P::VB + chn * 16) = ub[r];
  }
#pragma unroll
  for (int r = 0; r < P::RC; ++r) {
    const int i = r * 64 + lane;
    if (i < P::NC) *reinterpret_cast<uint4*>(Cn + i * 16) = cb[r];      // [col][oz][chunk] is the lane-linear order
  }
  __syncthreads();                                                       // wp (the neighbourhoods are wave-private: LDS runs a wave's accesses in order)

  // ---- conv11 + skip + channel dot products of the nine output columns ----
  const int lrc = lr < Dq ? lr : 0;                                      // lanes beyond the last input depth repeat voxel 0 (their results are not used)
  const unsigned bo = (unsigned)((lrc + (lg >> 1)) * P::VB + (lg & 1) * (P::VB / 2));      // B operand: u9(iz = lr + (lg >> 1)), channels 8 (lg & 1) ..
  const int oz_l = 2 * lrc + (lg >> 1);                                  // this lane's output depth: rows 0-7 even, rows 8-15 odd
  const unsigned co = (unsigned)(oz_l * P::CB + (lg & 1) * (P::CB / 2)); // c0 channels 4 (lg & 1) .. of that voxel
  unsigned colmask = 0;
#pragma unroll
  for (int c = 0; c < 9; ++c)
    colmask |= (unsigned)(active && (unsigned)(y - 1 + c / 3) < (unsigned)H && (unsigned)(x - 1 + c % 3) < (unsigned)W) << c;
  float b11[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) b11[e] = d.bias11[(lg & 1) * 4 + e];
  float S[3] = {0.f, 0.f, 0.f};
  const uint4* wq = reinterpret_cast<const uint4*>(d.w11);
  const int par = ((y & 1) << 1) | (x & 1);                              // wave-uniform
  if (par == 0) ps2_columns<T, 0, 0>(U, Cn, wp, wq, colmask, lr, lg, bo, co, b11, S);
  else if (par == 1) ps2_columns<T, 0, 1>(U, Cn, wp, wq, colmask, lr, lg, bo, co, b11, S);
  else if (par == 2) ps2_columns<T, 1, 0>(U, Cn, wp, wq, colmask, lr, lg, bo, co, b11, S);
  else ps2_columns<T, 1, 1>(U, Cn, wp, wq, colmask, lr, lg, bo, co, b11, S);

  // ---- logits: the three running sums meet across depths (kd = 0 comes from depth z - 1, kd = 2 from z + 1), then softmax and depth ----
  if (lr < Dq && (lg & 1) == 0) {
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) Sw[kd * PS_DMAX + oz_l] = S[kd];
  }
  float logit = -INFINITY;
  if (lane < D) {
    float acc = 0.f;
    if (lane > 0) acc += Sw[lane - 1];
    acc += Sw[PS_DMAX + lane];
    if (lane + 1 < D) acc += Sw[2 * PS_DMAX + lane + 1];
    logit = acc;
  }
  float m = logit;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  // a NaN logit must poison the whole pixel like torch.softmax does (fmaxf drops NaNs)
  float nanflag = (lane < D && logit != logit) ? 1.f : 0.f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) nanflag += __shfl_xor(nanflag, off);
  float e = lane < D ? expf(logit - m) : 0.f;
  float sum = e;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  float pr = e * (1.f / sum);
  if (nanflag > 0.f) pr = __builtin_nanf("");
  float dep = lane < D ? pr * d.depths[(v % d.B) * D + lane] : 0.f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) dep += __shfl_xor(dep, off);
  if (active) {
    if (lane < D) d.prob[o * D + lane] = pr;
    if (lane == 0) d.depth_out[o] = dep;
  }
}

// conv11's weights [16 in][8 out][3][3][3] (x folded BN scale) as the nine A operands of prob_sparse2_kernel, [9 in-plane taps kh * 3 + kw][16 rows]
// [4 K groups][8]: K = 32 = [u9(iz): 16 channels | u9(iz + 1): 16 channels]; rows 0-7 (even output depths) = W(kd = 1) on the first half, rows
// 8-15 (odd output depths) = W(kd = 2) on the first half and W(kd = 0) on the second.
void prob_sparse_pack(const float* w, const float* scale, std::vector<float>& packed) {
  packed.assign((size_t)9 * 16 * 4 * 8, 0.f);
  for (int t = 0; t < 9; ++t)
    for (int row = 0; row < 16; ++row)
      for (int k = 0; k < 32; ++k) {
        const int co = row & 7, ci = k & 15;
        int kd;
        if (row < 8) { if (k >= 16) continue; kd = 1; }
        else kd = k < 16 ? 2 : 0;
        packed[(((size_t)t * 16 + row) * 4 + (k >> 3)) * 8 + (k & 7)] = w[((long long)ci * 8 + co) * 27 + kd * 9 + t] * (scale ? scale[co] : 1.f);
      }
}

// ---------------------------------------------------------------------------------------------------------
// Sparse cost regularisation.  With the sparse tail the network reads the probability volume only at the chosen pixels, so every
// tensor of the 3-D U-Net is needed only inside the dependency cone of those pixels: per axis and per chosen coordinate y the
// needed index interval of each tensor follows from the layers' geometry (network_v5.py:260-291; Conv3d k3 p1 stride 1: i-1..i+1,
// stride 2: 2i-1..2i+1; ConvTranspose3d k3 s2 p1 op1: output o reads inputs floor(o/2)..ceil(o/2); skips read their own index):
//     u11 [y-1, y+1] -> u9 -> u7 -> c6 -> c5 -> c4 (input of conv5, and skip of u7) -> c3 -> c2 (input of conv3, skip of u9) -> c1
//     -> c0 (input of conv1, skip of u11): +-29 pixels around a chosen pixel at full resolution.
// A point's needed set is the product of its row and column intervals; a tile of a layer is needed if it intersects the box of
// any point of its view (all depths are needed).  Unneeded tiles are never computed: what is read later never depends on them
// (convolutions are local), so the numbers that reach the outputs are unchanged bit for bit.  One workgroup per view builds the
// eight tile masks in LDS; a second kernel compacts the depth-sweeping conv0's needed tiles into a list (its kernels walk it).
namespace {
struct Ival { int a, b; };
__device__ __forceinline__ Ival iv_clamp(int a, int b, int n) { return Ival{a < 0 ? 0 : a, b > n - 1 ? n - 1 : b}; }
__device__ __forceinline__ Ival iv_tr(Ival o, int n_in) { return iv_clamp(o.a >> 1, (o.b + 1) >> 1, n_in); }      // transposed conv: inputs of outputs [a, b]
__device__ __forceinline__ Ival iv_s1(Ival o, int n_in) { return iv_clamp(o.a - 1, o.b + 1, n_in); }
__device__ __forceinline__ Ival iv_s2(Ival o, int n_in) { return iv_clamp(2 * o.a - 1, 2 * o.b + 1, n_in); }
__device__ __forceinline__ Ival iv_or(Ival p, Ival q) { return Ival{p.a < q.a ? p.a : q.a, p.b > q.b ? p.b : q.b}; }
struct Cone { Ival c0, c1, c2, c3, c4, c5, u7, u9; };
__device__ __forceinline__ Cone cone_of(int y, int S) {
  Cone k;
  const Ival u11 = iv_clamp(y - 1, y + 1, S);
  k.u9 = iv_tr(u11, S / 2);
  k.u7 = iv_tr(k.u9, S / 4);
  const Ival c6 = iv_tr(k.u7, S / 8);
  k.c5 = iv_s1(c6, S / 8);
  k.c4 = iv_or(iv_s2(k.c5, S / 4), k.u7);
  k.c3 = iv_s1(k.c4, S / 4);
  k.c2 = iv_or(iv_s2(k.c3, S / 2), k.u9);
  k.c1 = iv_s1(k.c2, S / 2);
  k.c0 = iv_or(iv_s2(k.c1, S), u11);
  return k;
}
__device__ __forceinline__ void mark(unsigned char* m, int nw, Ival r, Ival c, int shr_r, int div_r, int shr_c, int div_c) {
  // tiles = index / div (div > 0) or index >> shr
  const int ra = div_r ? r.a / div_r : r.a >> shr_r, rb = div_r ? r.b / div_r : r.b >> shr_r;
  const int ca = div_c ? c.a / div_c : c.a >> shr_c, cb = div_c ? c.b / div_c : c.b >> shr_c;
  for (int i = ra; i <= rb; ++i)
    for (int j = ca; j <= cb; ++j) m[i * nw + j] = 1;
}
}  // namespace

// mask layout per view (bytes, row-major [rows][cols] of each layer's tile grid; S = crop size, 8 x 8-cell tiles for the halo-tile
// kernels, 12 x 16 pixels for the depth-sweeping conv0): offsets in SparseMaskLayout
struct SparseMaskLayout {
  int n0h, n0w, n1, n3, n5;      // tile grids: sweep n0h x n0w; conv1 / conv2 n1 x n1 (half resolution); conv3 / conv4 / conv9 n3 x n3; conv5 / conv7 n5 x n5
  int o0, o1, o2, o3, o4, o5, o7, o9, total;
};
static SparseMaskLayout sparse_mask_layout(int S) {
  SparseMaskLayout L;
  L.n0h = (S + 11) / 12; L.n0w = (S + 15) / 16; L.n1 = (S / 2 + 7) / 8; L.n3 = (S / 4 + 7) / 8; L.n5 = (S / 8 + 7) / 8;
  int o = 0;
  L.o0 = o; o += L.n0h * L.n0w;
  L.o1 = o; o += L.n1 * L.n1;
  L.o2 = o; o += L.n1 * L.n1;
  L.o3 = o; o += L.n3 * L.n3;
  L.o4 = o; o += L.n3 * L.n3;
  L.o5 = o; o += L.n5 * L.n5;
  L.o7 = o; o += L.n5 * L.n5;
  L.o9 = o; o += L.n3 * L.n3;
  L.total = (o + 15) & ~15;
  return L;
}
int sparse_mask_bytes_per_view(int S) { return sparse_mask_layout(S).total; }
int sparse_mask_offset(int S, int layer) {      // layer: 0 = sweep, 1..5 = conv1..conv5, 7 = conv7, 8 = conv9 (launch_conv3d_tile's ids)
  const SparseMaskLayout L = sparse_mask_layout(S);
  switch (layer) { case 0: return L.o0; case 1: return L.o1; case 2: return L.o2; case 3: return L.o3; case 4: return L.o4;
                   case 5: return L.o5; case 7: return L.o7; case 8: return L.o9; default: return -1; }
}

__global__ __launch_bounds__(1024) void sparse_mask_kernel(const int* __restrict__ choose, int v0, int P, int S, SparseMaskLayout L,
                                                          unsigned char* __restrict__ masks, int* __restrict__ view_count) {
  // one point per thread (1024 threads): a point marks up to a few dozen tiles byte by byte; with four points per thread the launch was
  // 20 us of a 1.26 ms forward at B = 1
  __shared__ unsigned char m[2048];
  const int vl = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < L.total; i += 1024) m[i] = 0;
  __syncthreads();
  for (int p = tid; p < P; p += 1024) {
    const int pix = choose[(long long)(v0 + vl) * P + p];
    const int y = pix / S, x = pix - y * S;
    const Cone r = cone_of(y, S), c = cone_of(x, S);
    mark(m + L.o0, L.n0w, r.c0, c.c0, 0, 12, 0, 16);
    mark(m + L.o1, L.n1, r.c1, c.c1, 3, 0, 3, 0);
    mark(m + L.o2, L.n1, r.c2, c.c2, 3, 0, 3, 0);
    mark(m + L.o3, L.n3, r.c3, c.c3, 3, 0, 3, 0);
    mark(m + L.o4, L.n3, r.c4, c.c4, 3, 0, 3, 0);
    mark(m + L.o5, L.n5, r.c5, c.c5, 3, 0, 3, 0);
    mark(m + L.o7, L.n5, r.u7, c.u7, 4, 0, 4, 0);      // transposed: the tile grid is the INPUT grid, two output cells per input cell
    mark(m + L.o9, L.n3, r.u9, c.u9, 4, 0, 4, 0);
  }
  __syncthreads();
  for (int i = tid; i < L.total; i += 1024) masks[(long long)vl * L.total + i] = m[i];
  // this view's number of sweep tiles (the list kernel's offsets)
  int c = 0;
  for (int i = tid; i < L.n0h * L.n0w; i += 1024) c += m[L.o0 + i];
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  __shared__ int wsum[16];
  if ((tid & 63) == 0) wsum[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int q = 0; q < 16; ++q) t += wsum[q];
    view_count[1 + vl] = t;
  }
}

// needed tiles of the depth-sweeping conv0 in ascending order: list[i] = (view * n0h + row tile) * n0w + column tile, count[0] = how many.
// One workgroup per view: its offset is the sum of the per-view counts (count[1 + u], written by sparse_mask_kernel) of the views before it.
__global__ __launch_bounds__(256) void sparse_sweep_list_kernel(const unsigned char* __restrict__ masks, int Vc, SparseMaskLayout L,
                                                                int* __restrict__ list, int* __restrict__ count) {
  __shared__ int part[256];
  const int per_view = L.n0h * L.n0w, v = blockIdx.x, tid = threadIdx.x;
  int c = 0;
  for (int u = tid; u < v; u += 256) c += count[1 + u];
  part[tid] = c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) part[tid] += part[tid + off];
    __syncthreads();
  }
  int base = part[0];
  __syncthreads();
  const unsigned char* m = masks + (long long)v * L.total + L.o0;
  for (int seg = 0; seg < per_view; seg += 256) {
    const int i = seg + tid;
    const int f = i < per_view && m[i] ? 1 : 0;
    part[tid] = f;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {      // inclusive scan
      const int t = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    if (f) list[base + part[tid] - 1] = v * per_view + i;
    base += part[255];
    __syncthreads();
  }
  if (v == Vc - 1 && tid == 0) count[0] = base;
}

// masks: Vc * sparse_mask_bytes_per_view(S) bytes; sweep_list: Vc * ceil(S/12) * ceil(S/16) ints; sweep_count: 1 + Vc ints (total, then per view)
int launch_sparse_masks(const int* choose, int v0, int Vc, int P, int S, unsigned char* masks, int* sweep_list, int* sweep_count,
                        hipStream_t s) {
  RGBM_REQUIRE(choose && masks && sweep_list && sweep_count && S % 8 == 0 && S >= 16 && Vc > 0, "sparse masks arguments");
  const SparseMaskLayout L = sparse_mask_layout(S);
  RGBM_REQUIRE(L.total <= 2048, "sparse masks: crop too large for the mask kernel's LDS table");
  hipLaunchKernelGGL(sparse_mask_kernel, dim3((unsigned)Vc), dim3(1024), 0, s, choose, v0, P, S, L, masks, sweep_count);
  hipLaunchKernelGGL(sparse_sweep_list_kernel, dim3((unsigned)Vc), dim3(256), 0, s, masks, Vc, L, sweep_list, sweep_count);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_prob_sparse(const void* u9, const void* c0, const void* w11_packed, const float* bias11, const float* wprob,
                       const int* choose, const float* depths, float* prob, float* depth_out, int v0, int Vc, int B, int P,
                       int D, int H, int W, int dtype, hipStream_t s, const void* w11_taps) {
  RGBM_REQUIRE(dtype == BF16 || dtype == F16 || dtype == BF16X3, "prob_sparse: 16-bit storage types or bf16x3");
  RGBM_REQUIRE(u9 && c0 && w11_packed && bias11 && wprob && choose && depths && prob && depth_out, "prob_sparse arguments");
  RGBM_REQUIRE(D <= PS_DMAX && (D % 2) == 0 && (H % 2) == 0 && (W % 2) == 0, "prob_sparse supports even D <= 24 and even H, W");
  RGBM_REQUIRE((long long)D * H * W * 8 < (1ll << 31), "prob_sparse view too large for 32-bit offsets");
  ProbSparseDesc d;
  d.u9 = u9; d.c0 = c0; d.w11 = w11_packed;
  d.bias11 = bias11; d.wprob = wprob; d.choose = choose; d.depths = depths; d.prob = prob; d.depth_out = depth_out;
  d.v0 = v0; d.Vc = Vc; d.B = B; d.P = P; d.D = D; d.H = H; d.W = W;
  const long long npts = (long long)Vc * P;
  RGBM_REQUIRE(npts > 0 && (npts + 3) / 4 < (1ll << 31), "prob_sparse grid out of range");
  if (w11_taps != nullptr && !(g_debug_flags & (1 << 29))) {
    // the round-6 kernel (w11_taps: prob_sparse_pack's operands)
    RGBM_REQUIRE((long long)(D / 2) * (H / 2) * (W / 2) * 16 * 4 < (1ll << 31) && (long long)D * H * W * 8 * 4 < (1ll << 31), "prob_sparse view too large for 32-bit offsets");
    d.w11 = w11_taps;
    const unsigned grid = (unsigned)((npts + 3) / 4);
#define PS2_LAUNCH(TT)                                                                                              \
    do {                                                                                                            \
      constexpr int lds = PS2_WP_BYTES + 4 * Ps2<TT>::WAVE_BYTES;                                                   \
      if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(prob_sparse2_kernel<TT>), lds)) return rc;      \
      hipLaunchKernelGGL(prob_sparse2_kernel<TT>, dim3(grid), dim3(256), lds, s, d);                                \
    } while (0)
    if (dtype == BF16) PS2_LAUNCH(unsigned short);
    else if (dtype == BF16X3) PS2_LAUNCH(bx3_t);
    else PS2_LAUNCH(f16_t);
#undef PS2_LAUNCH
    RGBM_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (dtype == BF16) hipLaunchKernelGGL(prob_sparse_kernel<unsigned short>, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, s, d);
  else if (dtype == BF16X3) hipLaunchKernelGGL(prob_sparse_kernel<bx3_t>, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, s, d);
  else hipLaunchKernelGGL(prob_sparse_kernel<f16_t>, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, s, d);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
