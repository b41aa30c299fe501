// ResNet stem in one kernel: conv1 7x7 stride 2 pad 3 (3 -> 64, no bias, pspnet.py:37) -> ReLU -> max-pool 3x3 stride 2 pad 1
// (pspnet.py:39), from the two NCHW fp32 image batches straight to the pooled [V][S/4][S/4][64] tensor in the storage type.
// Replaces three launches (NCHW -> padded NHWC copy, the 7x7 conv on the generic implicit-GEMM tile, the pool) and the
// 112 x 112 x 64 tensor between them (0.82 GB written and read back per step in bf16).
//
// A workgroup owns an 8 x 8 tile of POOLED pixels = 17 x 17 conv outputs (one row / column of overlap with its neighbours,
// recomputed) = a 39 x 39 pixel input patch.  The patch is staged in LDS as [row][col][4 channels] (the 4th is zero), so the
// eight K values an MFMA lane needs — two horizontally adjacent taps x 4 channels — are 16 contiguous bytes: K is laid out per
// kernel row as 8 column slots x 4 channels = 32 (the 8th slot and the 4th channel carry zero weights), one kernel row per
// 16x16x32 step, seven steps.  Wave w multiplies output channels 16 w .. 16 w + 15 against all 19 sixteen-pixel tiles of the
// 289 conv outputs; then ReLU, the conv tile goes to LDS (over the patch) and 256 threads pool 64 pixels x 64 channels out of it.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace rgbm {

namespace {

constexpr int kPT = 8;                      // pooled tile width
constexpr int kCT = 2 * kPT + 1;            // conv tile width: 17
constexpr int kPC = 2 * (kCT - 1) + 7 + 1;  // patch columns: 40 (39 + one: the zero-weight 8th column slot of the last pixel reads it)
constexpr int kNB = 5;                      // pixel tiles per read-ahead batch
// PTY: pooled tile height — 8 for the 16-bit types (17 x 17 conv outputs, 19 pixel tiles), 4 for split pairs (9 x 17, 10 tiles: the
// conv tile in 4-byte slots is 79 KB of LDS at height 8, two workgroups per CU and 186 registers)
template <int PTY> struct StemGeo {
  static constexpr int CTY = 2 * PTY + 1;             // conv tile height
  static constexpr int PR = 2 * (CTY - 1) + 7;        // patch rows
  static constexpr int NPX = CTY * kCT;               // conv outputs per tile
  static constexpr int NTL = (NPX + 15) / 16;         // 16-pixel tiles
};

template <typename T> struct StemMma;
template <> struct StemMma<unsigned short> {
  __device__ static __forceinline__ f32x4 run(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct StemMma<f16_t> {
  __device__ static __forceinline__ f32x4 run(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

template <typename T, int PTY>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                   const T* __restrict__ wpk, T* __restrict__ out, int B, int V, int S) {
  constexpr bool X3 = std::is_same<T, bx3_t>::value;
  constexpr int kPR = StemGeo<PTY>::PR, kNPX = StemGeo<PTY>::NPX, kNTL = StemGeo<PTY>::NTL;
  constexpr int EB = (int)sizeof(T);
  constexpr int PXB = 4 * EB;                       // bytes per patch pixel (4 channels): 8 / 16
  constexpr int SROW = 64 * EB + 16;                // staging row (one conv output, 64 channels) + bank spread
  typedef typename std::conditional<X3, unsigned short, T>::type MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* patch = lds;                       // [39][40] pixels
  unsigned char* stg = lds;                         // [289][SROW] conv outputs (after the products; aliases the patch)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int Sc = S / 2, Sp = S / 4, ntx = Sp / kPT, nty = Sp / PTY;
  // every XCD (own L2) takes one contiguous eighth of the tiles: neighbours share their patch borders
  const unsigned ntile = (unsigned)(V * ntx * nty), per_xcd = (ntile + 7) / 8;
  const unsigned tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= ntile) return;
  const int v = (int)(tile / (unsigned)(ntx * nty));
  const int rem = (int)(tile - (unsigned)v * (unsigned)(ntx * nty));
  const int py0 = (rem / ntx) * PTY, px0 = (rem % ntx) * kPT;
  const float* img = v < B ? img1 + (long long)v * 3 * S * S : img2 + (long long)(v - B) * 3 * S * S;

  // ---- input patch: rows 4 py0 - 5 .., columns 4 px0 - 5 .. (zero outside the image: the conv's padding) ----
  for (int i = tid; i < kPR * kPC; i += 256) {
    const int r = i / kPC, c = i - r * kPC;
    const int gy = 4 * py0 - 5 + r, gx = 4 * px0 - 5 + c;
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S) {
      const float* q = img + (long long)gy * S + gx;
      p[0] = q[0]; p[1] = q[(long long)S * S]; p[2] = q[2ll * S * S];
    }
    store4(reinterpret_cast<T*>(patch + i * PXB), p);
  }

  // ---- this wave's weights: output channels 16 wv + lr, kernel row kh, K group lg (column slots 2 lg, 2 lg + 1 x 4 channels) ----
  uint4 wa[7][X3 ? 2 : 1];
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    const T* wp = wpk + ((kh * 64 + 16 * wv + lr) * 32 + lg * 8);
    wa[kh][0] = *reinterpret_cast<const uint4*>(wp);
    if constexpr (X3) wa[kh][1] = *reinterpret_cast<const uint4*>(wp + 4);
  }
  // per pixel tile: patch offset of the lane's conv output (row 2 cy, column 2 cx + 2 lg); outputs past 288 recompute 288
  int base[kNTL];
#pragma unroll
  for (int n = 0; n < kNTL; ++n) {
    const int p = min(n * 16 + lr, kNPX - 1);
    const int cy = p / kCT, cx = p - cy * kCT;
    base[n] = ((2 * cy) * kPC + 2 * cx + 2 * lg) * PXB;
  }
  f32x4 acc[kNTL];
#pragma unroll
  for (int n = 0; n < kNTL; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // ---- seven kernel rows x 19 pixel tiles; the B operands of the next batch of tiles are read before the MFMAs of this one ----
  constexpr int NBT = (kNTL + kNB - 1) / kNB;       // batches per kernel row
  uint4 bq[2][kNB][X3 ? 2 : 1];
  auto read_b = [&](int it, uint4 (&dst)[kNB][X3 ? 2 : 1]) {
    const int kh = it / NBT, n0 = (it - kh * NBT) * kNB;
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
      if (n0 + j < kNTL) {
        const unsigned char* q = patch + base[n0 + j] + kh * (kPC * PXB);
        dst[j][0] = *reinterpret_cast<const uint4*>(q);
        if constexpr (X3) dst[j][1] = *reinterpret_cast<const uint4*>(q + 16);
      }
    }
  };
  read_b(0, bq[0]);
#pragma unroll
  for (int it = 0; it < 7 * NBT; ++it) {
    const int kh = it / NBT, n0 = (it - kh * NBT) * kNB;
    if (it + 1 < 7 * NBT) read_b(it + 1, bq[(it + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (X3) {
      uint4 ah, al;
      bx3_pair(wa[kh][0], wa[kh][1], ah, al);
      uint4 bh[kNB], bl[kNB];
#pragma unroll
      for (int j = 0; j < kNB; ++j)
        if (n0 + j < kNTL) bx3_pair(bq[it & 1][j][0], bq[it & 1][j][1], bh[j], bl[j]);
      // small terms first (w_lo x_hi, w_hi x_lo, w_hi x_hi), every accumulator once per term
#pragma unroll
      for (int j = 0; j < kNB; ++j)
        if (n0 + j < kNTL) acc[n0 + j] = StemMma<MT>::run(al, bh[j], acc[n0 + j]);
#pragma unroll
      for (int j = 0; j < kNB; ++j)
        if (n0 + j < kNTL) acc[n0 + j] = StemMma<MT>::run(ah, bl[j], acc[n0 + j]);
#pragma unroll
      for (int j = 0; j < kNB; ++j)
        if (n0 + j < kNTL) acc[n0 + j] = StemMma<MT>::run(ah, bh[j], acc[n0 + j]);
    } else {
#pragma unroll
      for (int j = 0; j < kNB; ++j)
        if (n0 + j < kNTL) acc[n0 + j] = StemMma<MT>::run(wa[kh][0], bq[it & 1][j][0], acc[n0 + j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- ReLU, conv tile -> LDS (storage type: what the unfused path wrote to memory) ----
  __syncthreads();                                  // every wave has read its last patch operands
#pragma unroll
  for (int n = 0; n < kNTL; ++n) {
    const int p = n * 16 + lr;
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = acc[n][e] < 0.f ? 0.f : acc[n][e];        // NaN passes, like torch.relu
    if (p < kNPX) store4(reinterpret_cast<T*>(stg + p * SROW) + 16 * wv + 4 * lg, r);
  }
  __syncthreads();

  // ---- 3 x 3 stride-2 max over the conv tile: thread = (pooled pixel, 16 channels); NaN wins like torch.max_pool2d ----
  constexpr int E = 16 / EB;                        // channels per 16-byte chunk
  constexpr int NCH = 16 / E;                       // chunks per thread
  const int pp = tid >> 2, cg = tid & 3;
  if (pp >= PTY * kPT) return;                      // a 4-row tile has 32 pooled pixels: half of the threads are done
  const int py = pp >> 3, px = pp & 7;
  float m[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) m[e] = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int cr = 2 * (py0 + py) - 1 + dy;         // conv row in the image
    if ((unsigned)cr >= (unsigned)Sc) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int cc = 2 * (px0 + px) - 1 + dx;
      if ((unsigned)cc >= (unsigned)Sc) continue;
      const unsigned char* q = stg + ((2 * py + dy) * kCT + 2 * px + dx) * SROW + cg * 16 * EB;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        float xv[E];
        unpack_chunk(*reinterpret_cast<const uint4*>(q + k * 16), xv, T());
#pragma unroll
        for (int e = 0; e < E; ++e) m[k * E + e] = (xv[e] > m[k * E + e] || xv[e] != xv[e]) ? xv[e] : m[k * E + e];
      }
    }
  }
  T* o = out + (((long long)v * Sp + py0 + py) * Sp + px0 + px) * 64 + cg * 16;
#pragma unroll
  for (int k = 0; k < NCH; ++k) *reinterpret_cast<uint4*>(o + k * E) = pack_chunk(m + k * E, T());
}

template <typename T, int PTY>
int launch_t(const float* img1, const float* img2, const void* wpk, void* out, int B, int V, int S, hipStream_t s) {
  constexpr int EB = (int)sizeof(T);
  constexpr int lds_patch = StemGeo<PTY>::PR * kPC * 4 * EB, lds_stg = StemGeo<PTY>::NPX * (64 * EB + 16);
  constexpr int lds = lds_patch > lds_stg ? lds_patch : lds_stg;
  auto kern = stem_kernel<T, PTY>;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
  const int ntx = S / 4 / kPT, nty = S / 4 / PTY;
  const long long ntile = (long long)V * ntx * nty;
  RGBM_REQUIRE(ntile > 0 && ntile < (1ll << 30), "stem grid out of range");
  hipLaunchKernelGGL(kern, dim3((unsigned)(((ntile + 7) / 8) * 8)), dim3(256), lds, s, img1, img2, (const T*)wpk, (T*)out, B, V, S);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// w: conv1.weight [64][3][7][7] -> the kernel's A operands [kh][co][kw slot 0..7][c 0..3] (slot 7 and channel 3 zero), fp32 values
void stem_pack(const float* w, std::vector<float>& packed) {
  packed.assign((size_t)7 * 64 * 32, 0.f);
  for (int kh = 0; kh < 7; ++kh)
    for (int co = 0; co < 64; ++co)
      for (int kw = 0; kw < 7; ++kw)
        for (int c = 0; c < 3; ++c) packed[((size_t)(kh * 64 + co)) * 32 + kw * 4 + c] = w[((co * 3 + c) * 7 + kh) * 7 + kw];
}

// img1 / img2 [B][3][S][S] fp32 (views 0..B-1 / B..2B-1) -> out [V][S/4][S/4][64] = maxpool3x3s2p1(relu(conv7x7s2p3(img)))
int launch_stem(int dtype, const float* img1, const float* img2, const void* wpk, void* out, int B, int V, int S, hipStream_t s) {
  RGBM_REQUIRE(dtype == BF16 || dtype == F16 || dtype == BF16X3, "stem kernel: 16-bit or split-pair storage");
  RGBM_REQUIRE(S % (4 * kPT) == 0 && V == 2 * B && img1 && img2 && wpk && out, "stem kernel geometry");
  if (dtype == BF16) return launch_t<unsigned short, 8>(img1, img2, wpk, out, B, V, S, s);
  if (dtype == F16) return launch_t<f16_t, 8>(img1, img2, wpk, out, B, V, S, s);
  return launch_t<bx3_t, 4>(img1, img2, wpk, out, B, V, S, s);
}

}  // namespace rgbm
