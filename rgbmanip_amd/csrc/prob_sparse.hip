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
  const long long pidx = (long long)blockIdx.x * 4 + wave;              // point index within the chunk (wave-uniform)
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
// Sparse decoder: with the sparse tail, u9 (conv9's output) is read only under the 3 x 3 neighbourhoods of the chosen pixels and
// u7 (conv7's) only under the conv9 tiles that cover those — the halo-tile kernels of both layers skip every (view, row tile,
// column tile) whose byte is 0 (Conv3dTileDesc::tile_mask; all depth tiles share it).  8 x 8 tiles of the transposed convs' INPUT
// grids: conv9 tile t covers half-resolution rows 16 t .. 16 t + 15, conv7 tile t quarter-resolution rows 16 t .. 16 t + 15.
__global__ __launch_bounds__(256) void decoder_tile_mask_kernel(const int* __restrict__ choose, int v0, int Vc, int P, int H, int W,
                                                                unsigned char* __restrict__ mask9, int nt9,
                                                                unsigned char* __restrict__ mask7, int nt7) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)Vc * P) return;
  const int vl = (int)(i / P);
  const int pix = choose[(long long)(v0 + vl) * P + (i - (long long)vl * P)];
  const int y = pix / W, x = pix - y * W;
  const int Hh = H >> 1, Wh = W >> 1, Hq = H >> 2, Wq = W >> 2;
  // half-resolution rows / columns of u9 that conv11 reads for full-resolution rows y - 1 .. y + 1 (prob_sparse_kernel: input voxel
  // (row >> 1) of tap 0 and the one after it)
  const int r0 = max(y - 1, 0) >> 1, r1 = min((min(y + 1, H - 1) >> 1) + 1, Hh - 1);
  const int c0 = max(x - 1, 0) >> 1, c1 = min((min(x + 1, W - 1) >> 1) + 1, Wh - 1);
  for (int a = r0 >> 4; a <= r1 >> 4; ++a)
    for (int b = c0 >> 4; b <= c1 >> 4; ++b) {
      mask9[((long long)vl * nt9 + a) * nt9 + b] = 1;
      // conv9 tile (a, b) reads quarter-resolution rows 8 a .. 8 a + 8 of u7 (its 9-row halo) = conv7 tiles (8 a) >> 4 .. (8 a + 8) >> 4
      const int q0 = 8 * a, q1 = min(8 * a + 8, Hq - 1), p0 = 8 * b, p1 = min(8 * b + 8, Wq - 1);
      for (int e = q0 >> 4; e <= q1 >> 4; ++e)
        for (int f = p0 >> 4; f <= p1 >> 4; ++f) mask7[((long long)vl * nt7 + e) * nt7 + f] = 1;
    }
}

// mask9 [Vc][nt9][nt9], mask7 [Vc][nt7][nt7] bytes (nt9 = ceil(W / 4 / 8), nt7 = ceil(W / 8 / 8)); H == W
int launch_decoder_tile_masks(const int* choose, int v0, int Vc, int P, int H, int W, unsigned char* mask9, unsigned char* mask7,
                              hipStream_t s) {
  RGBM_REQUIRE(choose && mask9 && mask7 && H == W && (H % 8) == 0, "decoder tile masks arguments");
  const int nt9 = (W / 4 + 7) / 8, nt7 = (W / 8 + 7) / 8;
  RGBM_CHECK_HIP(hipMemsetAsync(mask9, 0, (size_t)Vc * nt9 * nt9, s));
  RGBM_CHECK_HIP(hipMemsetAsync(mask7, 0, (size_t)Vc * nt7 * nt7, s));
  const long long n = (long long)Vc * P;
  hipLaunchKernelGGL(decoder_tile_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, choose, v0, Vc, P, H, W, mask9, nt9, mask7, nt7);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_prob_sparse(const void* u9, const void* c0, const void* w11_packed, const float* bias11, const float* wprob,
                       const int* choose, const float* depths, float* prob, float* depth_out, int v0, int Vc, int B, int P,
                       int D, int H, int W, int dtype, hipStream_t s) {
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
  if (dtype == BF16) hipLaunchKernelGGL(prob_sparse_kernel<unsigned short>, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, s, d);
  else if (dtype == BF16X3) hipLaunchKernelGGL(prob_sparse_kernel<bx3_t>, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, s, d);
  else hipLaunchKernelGGL(prob_sparse_kernel<f16_t>, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, s, d);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
