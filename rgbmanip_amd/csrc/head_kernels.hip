// Point-sampled head kernels of the AdaPose forward (network_v5.py:432-508), gfx950.
// Everything here is fp32 (fp32 from the probability volume onward); feature maps /
// cost-volume activations may be bf16 or f32 (template T).  The pose MLP's per-point layers run
// through the implicit-GEMM kernels as 1x1 convolutions; the per-point NOCS branch is one kernel here
// (point_mlp_kernel, round 6).  The rest is the glue: gathers at `choose`, the sparse probability
// conv + softmax + depth regression, the depth-guided fusion (re-warping only the 1024x24 samples that
// are consumed), reductions and the tiny per-view regressors + Ortho6d.
#include "common.h"
#include "kernels.h"

namespace rgbm {

// ---------------------------------------------------------------- gather feat at choose -> fp32 rows
template <typename T>
__global__ void gather_points_kernel(const T* __restrict__ feat, const int* __restrict__ choose, float* __restrict__ out,
                                     int V, int P, int HW, int C) {
  // one thread per (point, 4 channels)
  const int q = C / 4;
  const long long total = (long long)V * P * q;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % q);
    const long long vp = i / q;
    const long long v = vp / P;
    const int pix = choose[vp];
    float x[4];
    load4(feat + (v * HW + pix) * C + c4 * 4, x);
    store4(out + vp * C + c4 * 4, x);
  }
}

int launch_gather_points(int dtype, const void* feat, const int* choose, float* out, int V, int P, int HW, int C,
                         hipStream_t s) {
  const long long total = (long long)V * P * (C / 4);
  const unsigned g = (unsigned)((total + 255) / 256);
  if (dtype == BF16)
    hipLaunchKernelGGL(gather_points_kernel<unsigned short>, dim3(g), dim3(256), 0, s, (const unsigned short*)feat, choose,
                       out, V, P, HW, C);
  else if (dtype == F16)
    hipLaunchKernelGGL(gather_points_kernel<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)feat, choose,
                       out, V, P, HW, C);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(gather_points_kernel<bx3_t>, dim3(g), dim3(256), 0, s, (const bx3_t*)feat, choose,
                       out, V, P, HW, C);
  else
    hipLaunchKernelGGL(gather_points_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)feat, choose, out, V, P, HW, C);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- the per-point NOCS branch in one kernel (network_v5.py:432-444)
// gather feat at choose -> instance_color (32 -> 64, ReLU) -> nocs_head (64 -> 128 ReLU -> 64 ReLU -> 3 Tanh) -> nocs_pts_mlp (3 -> 32 ReLU
// -> 64 ReLU, written into channels 32..95 of the pose feature).  Rounds 1-5 ran it as a gather launch and six fp32 implicit-GEMM launches
// of 32..128 channels: 65 us of a 1.2 ms forward at B = 1 and 0.55 ms of a batch-256 step at a few per cent of the fp32 matrix pipe, the
// five intermediate tensors (1.2 GB at batch 256) written and read back.  Here a wave carries 16 points through all six layers: the
// weights (88 KB of fp32, in MFMA-fragment order: one ds_read_b128 per lane feeds four v_mfma_f32_16x16x4_f32) and a wave's two activation
// buffers ([channel][16 points], so that a B operand is one contiguous ds_read_b32 per K step) live in LDS; four output-channel tiles
// accumulate side by side (independent accumulators hide the MFMA's latency behind the one wave per SIMD); bias and activation as in the
// GEMM epilogue (fp32 sums over ascending K, bias added behind them).  One workgroup of four waves per CU walks 64-point tiles.
// Layers 3 / 4 (64 -> 3 and 3 -> 32) run on a 16-channel tile with zero weights beyond channel 2: tanh(0) = 0 there, and 0 * 0 adds nothing.

namespace pmlp {
constexpr int NL = 6;
constexpr int cin(int l) { return l == 0 ? 32 : l == 1 ? 64 : l == 2 ? 128 : l == 3 ? 64 : l == 4 ? 16 : 32; }      // K of a layer's MFMA chain (layer 4: 3 real + 13 zero)
constexpr int cout(int l) { return l == 0 ? 64 : l == 1 ? 128 : l == 2 ? 64 : l == 3 ? 16 : l == 4 ? 32 : 64; }     // (layer 3: 3 real + 13 zero)
constexpr int act(int l) { return l == 3 ? ACT_TANH : ACT_RELU; }
constexpr int w_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += cin(i) * cout(i); return o; }
constexpr int b_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += cout(i); return o; }
constexpr int W_FLOATS = w_off(NL), B_FLOATS = b_off(NL);
constexpr int ACT_A = 128 * 16, ACT_B = 64 * 16;          // floats of a wave's two activation buffers (layer inputs 0 / 2 / 4 in A, 1 / 3 / 5 in B)
constexpr size_t LDS_BYTES = (size_t)(W_FLOATS + B_FLOATS + 4 * (ACT_A + ACT_B)) * sizeof(float);
}  // namespace pmlp

template <int L>
__device__ __forceinline__ void pmlp_layer(const float* __restrict__ wl, const float* __restrict__ bl, const float* __restrict__ xin,
                                           float* __restrict__ xout, int lane, float* __restrict__ gout4, float* __restrict__ gout64, int ldg) {
  using namespace pmlp;
  constexpr int CI = cin(L), CO = cout(L), KS4 = CI / 16, NT = CO / 16, G = NT < 4 ? NT : 4, AC = act(L);
  const float* wf = wl + w_off(L);
  const float* bf = bl + b_off(L);
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += G) {
    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k4 = 0; k4 < KS4; ++k4) {
      float bx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bx[j] = xin[(k4 * 4 + j) * 64 + lane];                      // X[k = 16 k4 + 4 j + lane / 16][point lane % 16]
      f32x4 a[G];
#pragma unroll
      for (int g = 0; g < G; ++g) a[g] = *reinterpret_cast<const f32x4*>(wf + (((t0 + g) * KS4 + k4) * 64 + lane) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][j], bx[j], acc[g], 0, 0, 0);
    }
    // rows 4 (lane / 16) + i of tile t0 + g, point lane % 16
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int ch = (t0 + g) * 16 + 4 * (lane >> 4);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bf + ch);
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r = acc[g][i] + bv[i];
        v[i] = AC == ACT_RELU ? (r < 0.f ? 0.f : r) : AC == ACT_TANH ? tanhf(r) : r;      // NaN propagates like torch
      }
      if (L == 3) {
        if (lane < 16) *reinterpret_cast<float4*>(gout4 + (long long)lane * 4) = make_float4(v[0], v[1], v[2], v[3]);      // nocs4: channels 0..3 of the tile
      }
      if (L == NL - 1) {
        *reinterpret_cast<float4*>(gout64 + (long long)(lane & 15) * ldg + ch) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) xout[(ch + i) * 16 + (lane & 15)] = v[i];
      }
    }
  }
}

template <typename TF>
__global__ __launch_bounds__(256) void point_mlp_kernel(const PointMlpDesc d) {
  using namespace pmlp;
  extern __shared__ __attribute__((aligned(16))) float pm_lds[];
  float* wl = pm_lds;
  float* bl = wl + W_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* xa = bl + B_FLOATS + wave * (ACT_A + ACT_B);
  float* xb = xa + ACT_A;
  // the table (weights in MFMA-fragment order, then the biases: point_mlp_pack) as it lies in memory: 16-byte copies, all of a thread's in flight
  for (int i = tid; i < (W_FLOATS + B_FLOATS) / 4; i += 256) reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(d.table)[i];
  __syncthreads();
  const TF* __restrict__ feat = reinterpret_cast<const TF*>(d.feat);
  const long long ntile = d.N / 64;
  for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long long p0 = tile * 64 + wave * 16;            // this wave's 16 points
    // gather: 16 points x 8 four-channel groups = 128 loads, two per lane
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int q = lane + 64 * r, pt = q >> 3, c4 = q & 7;
      const long long vp = p0 + pt;
      const long long v = vp / d.P;
      const int pix = d.choose[vp];
      float x[4];
      load4(feat + (v * d.HW + pix) * 32 + c4 * 4, x);
#pragma unroll
      for (int e = 0; e < 4; ++e) xa[(c4 * 4 + e) * 16 + pt] = x[e];
    }
    // (wave-private buffers: a wave's LDS writes are seen by its own later reads in program order, no barrier)
    pmlp_layer<0>(wl, bl, xa, xb, lane, nullptr, nullptr, 0);
    pmlp_layer<1>(wl, bl, xb, xa, lane, nullptr, nullptr, 0);
    pmlp_layer<2>(wl, bl, xa, xb, lane, nullptr, nullptr, 0);
    pmlp_layer<3>(wl, bl, xb, xa, lane, d.nocs4 + p0 * 4, nullptr, 0);
    pmlp_layer<4>(wl, bl, xa, xb, lane, nullptr, nullptr, 0);
    pmlp_layer<5>(wl, bl, xb, xa, lane, nullptr, d.pf + p0 * d.ldpf, d.ldpf);
  }
}

int point_mlp_table_floats() { return pmlp::W_FLOATS + pmlp::B_FLOATS; }

// host: the six layers' weights ([Cout][Cin] row-major, as the state dict holds them) and biases into the kernel's LDS image —
// slot w_off(l) + ((t * KS4 + k4) * 64 + lane) * 4 + j = W_l[16 t + lane % 16][16 k4 + 4 j + lane / 16], zero outside the layer's shape
void point_mlp_pack(const float* const w[6], const float* const b[6], float* table) {
  using namespace pmlp;
  const int cin_real[NL] = {32, 64, 128, 64, 3, 32}, cout_real[NL] = {64, 128, 64, 3, 32, 64};
  for (int l = 0; l < NL; ++l) {
    const int CI = cin(l), CO = cout(l), KS4 = CI / 16;
    for (int i = 0; i < CI * CO; ++i) {
      const int j = i & 3, ln = (i >> 2) & 63, q = i >> 8;
      const int k4 = q % KS4, t = q / KS4;
      const int row = 16 * t + (ln & 15), k = 16 * k4 + 4 * j + (ln >> 4);
      table[w_off(l) + i] = (row < cout_real[l] && k < cin_real[l]) ? w[l][(size_t)row * cin_real[l] + k] : 0.f;
    }
    for (int i = 0; i < CO; ++i) table[W_FLOATS + b_off(l) + i] = (b[l] != nullptr && i < cout_real[l]) ? b[l][i] : 0.f;
  }
}

int launch_point_mlp(int feat_dtype, const PointMlpDesc& d, hipStream_t s) {
  RGBM_REQUIRE(d.N > 0 && d.N % 64 == 0 && d.P > 0 && d.ldpf % 4 == 0 && d.table, "point MLP: V * P must be a multiple of 64");
  static_assert((pmlp::W_FLOATS + pmlp::B_FLOATS) % 4 == 0, "the table is copied in 16-byte pieces");
  static_assert(pmlp::LDS_BYTES <= 160 * 1024, "point MLP: weights + four waves' activations must fit the LDS");
  int n_cu = 0;
  if (int rc = persistent_grid_cus(&n_cu)) return rc;
  const long long ntile = d.N / 64;
  const unsigned grid = (unsigned)(ntile < n_cu ? ntile : n_cu);
  const void* fn = feat_dtype == BF16 ? reinterpret_cast<const void*>(point_mlp_kernel<unsigned short>)
                   : feat_dtype == F16 ? reinterpret_cast<const void*>(point_mlp_kernel<f16_t>)
                   : feat_dtype == BF16X3 ? reinterpret_cast<const void*>(point_mlp_kernel<bx3_t>)
                                          : reinterpret_cast<const void*>(point_mlp_kernel<float>);
  if (int rc = ensure_dynamic_lds(fn, (int)pmlp::LDS_BYTES)) return rc;
  if (feat_dtype == BF16)
    hipLaunchKernelGGL(point_mlp_kernel<unsigned short>, dim3(grid), dim3(256), pmlp::LDS_BYTES, s, d);
  else if (feat_dtype == F16)
    hipLaunchKernelGGL(point_mlp_kernel<f16_t>, dim3(grid), dim3(256), pmlp::LDS_BYTES, s, d);
  else if (feat_dtype == BF16X3)
    hipLaunchKernelGGL(point_mlp_kernel<bx3_t>, dim3(grid), dim3(256), pmlp::LDS_BYTES, s, d);
  else
    hipLaunchKernelGGL(point_mlp_kernel<float>, dim3(grid), dim3(256), pmlp::LDS_BYTES, s, d);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- sparse prob conv + softmax + depth
// prob = Conv3d(8->1, 3^3, pad 1, no bias) evaluated only at the P chosen pixels x D depths
// (network_v5.py:280,290,449-455); softmax over depth; depth = sum p*d (network_v5.py:293-299).
// block = 256 threads = 64 points x 4 depth groups.
template <typename T>
__global__ __launch_bounds__(256) void prob_softmax_depth_kernel(const T* __restrict__ u11, const float* __restrict__ wprob,
                                                                 const int* __restrict__ choose,
                                                                 const float* __restrict__ depths, float* __restrict__ prob,
                                                                 float* __restrict__ depth_out, int v0, int Vc, int B, int P,
                                                                 int D, int H, int W, int classmajor) {
  constexpr int C = 8;
  __shared__ float w[27 * C];
  __shared__ float logit[64][25];
  for (int i = threadIdx.x; i < 27 * C; i += 256) w[i] = wprob[i];
  __syncthreads();
  const int pl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long pidx = (long long)blockIdx.x * 64 + pl;      // point index within the chunk
  const bool active = pidx < (long long)Vc * P;
  const int vl = active ? (int)(pidx / P) : 0;                 // local view
  const int v = v0 + vl;
  const int pix = active ? choose[(long long)v * P + (pidx - (long long)vl * P)] : 0;
  const int y = pix / W, x = pix - y * W;
  const int dper = (D + 3) / 4;
  if (active) {
    for (int dz = g * dper; dz < D && dz < (g + 1) * dper; ++dz) {
      float acc = 0.f;
      for (int kd = 0; kd < 3; ++kd) {
        const int zz = dz + kd - 1;
        if ((unsigned)zz >= (unsigned)D) continue;
        for (int kh = 0; kh < 3; ++kh) {
          const int yy = y + kh - 1;
          if ((unsigned)yy >= (unsigned)H) continue;
          for (int kw = 0; kw < 3; ++kw) {
            const int xx = x + kw - 1;
            if ((unsigned)xx >= (unsigned)W) continue;
            // class-major u11 (written by the halo-tiled conv11): class = parity bits (d,y,x), dense [8][Vc][D/2][H/2][W/2][C]
            const long long vidx = classmajor
                ? (((((long long)(((zz & 1) << 2) | ((yy & 1) << 1) | (xx & 1)) * Vc + vl) * (D >> 1) + (zz >> 1)) * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1))
                : ((((long long)vl * D + zz) * H + yy) * W + xx);
            const T* p = u11 + vidx * C;
            float a[4], b[4];
            load4(p, a);
            load4(p + 4, b);
            const float* ww = w + ((kd * 3 + kh) * 3 + kw) * C;
            acc += a[0] * ww[0] + a[1] * ww[1] + a[2] * ww[2] + a[3] * ww[3] + b[0] * ww[4] + b[1] * ww[5] + b[2] * ww[6] +
                   b[3] * ww[7];
          }
        }
      }
      logit[pl][dz] = acc;
    }
  }
  __syncthreads();
  if (active && g == 0) {
    float m = -INFINITY;
    for (int dz = 0; dz < D; ++dz) m = fmaxf(m, logit[pl][dz]);
    float sum = 0.f;
    for (int dz = 0; dz < D; ++dz) { const float e = expf(logit[pl][dz] - m); logit[pl][dz] = e; sum += e; }
    const float inv = 1.f / sum;
    float dep = 0.f;
    const int b = v % B;
    const long long o = ((long long)v * P + (pidx - (long long)vl * P));
    for (int dz = 0; dz < D; ++dz) {
      const float pr = logit[pl][dz] * inv;
      prob[o * D + dz] = pr;
      dep += pr * depths[b * D + dz];
    }
    depth_out[o] = dep;
  }
}

int launch_prob_softmax_depth(int dtype, const void* u11, const float* wprob, const int* choose, const float* depths,
                              float* prob, float* depth_out, int v0, int Vc, int B, int P, int D, int H, int W,
                              int classmajor, hipStream_t s) {
  RGBM_REQUIRE(D <= 24, "prob kernel supports up to 24 depth planes");
  const unsigned g = (unsigned)(((long long)Vc * P + 63) / 64);
  if (dtype == BF16)
    hipLaunchKernelGGL(prob_softmax_depth_kernel<unsigned short>, dim3(g), dim3(256), 0, s, (const unsigned short*)u11, wprob,
                       choose, depths, prob, depth_out, v0, Vc, B, P, D, H, W, classmajor);
  else if (dtype == F16)
    hipLaunchKernelGGL(prob_softmax_depth_kernel<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)u11, wprob,
                       choose, depths, prob, depth_out, v0, Vc, B, P, D, H, W, classmajor);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(prob_softmax_depth_kernel<bx3_t>, dim3(g), dim3(256), 0, s, (const bx3_t*)u11, wprob,
                       choose, depths, prob, depth_out, v0, Vc, B, P, D, H, W, classmajor);
  else
    hipLaunchKernelGGL(prob_softmax_depth_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)u11, wprob, choose, depths,
                       prob, depth_out, v0, Vc, B, P, D, H, W, classmajor);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- depth-guided fusion (network_v5.py:457-465)
// fg[v,p,c] = sum_d prob[v,p,d] * (feat[v,pix,c] + warp(feat[partner], v, d, pix)[c]);  written into a
// row-major [V*P][ldo] fp32 matrix at channel offset ch_off (the pose_mlp1 input concat).
__device__ __forceinline__ void warp_coords_h(const float* __restrict__ hm, float x, float y, float depth, int H, int W,
                                              float& ix, float& iy) {
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

template <typename T> __device__ __forceinline__ float round_through(float f) { return f; }
template <> __device__ __forceinline__ float round_through<unsigned short>(float f) { return bf16_to_f32(f32_to_bf16(f)); }
template <> __device__ __forceinline__ float round_through<f16_t>(float f) { return (float)(f16_t)sat_f16(f); }

template <typename T, bool ROUND_BF16>
__global__ void fuse_points_kernel(const T* __restrict__ feat, const float* __restrict__ homog, const float* __restrict__ depths,
                                   const int* __restrict__ choose, const float* __restrict__ prob, float* __restrict__ out,
                                   int V, int B, int P, int D, int H, int W, int ldo, int ch_off, int Vn) {
  // one thread per (point, 4-channel group): 8 threads per point for C=32; views 0 .. Vn - 1 (their partners are any of the V)
  constexpr int C = 32;
  const long long total = (long long)Vn * P * (C / 4);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i & 7);
    const long long vp = i >> 3;
    const int v = (int)(vp / P);
    const int partner = (v + B) % V;
    const int b = v % B;
    const int pix = choose[vp];
    const int y = pix / W, x = pix - y * W;
    float ref[4];
    load4(feat + (((long long)v * H + y) * W + x) * C + c4 * 4, ref);
    const T* src = feat + (long long)partner * H * W * C + c4 * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // The 8 lanes of a point share the projection: lane j projects depths j, j+8, j+16, ... and the others fetch the
    // coordinates by shuffle.  Corner reads are unconditional (clamped address, weight 0 outside the image: grid_sample's
    // zero padding): four independent reads per depth instead of four dependent branches.
    const int lane = threadIdx.x & 63, grp = lane & ~7;
    float cix[3], ciy[3];                                   // D <= 24
    // The 64 lanes of a wave belong to one view (P is a multiple of 8): the view's homography comes through the scalar cache into SGPRs
    // (one s_load instead of three vector loads per lane).  Round 4: with per-lane vector loads of this record feeding hipcc's packed
    // fp32 instructions, this kernel's warp coordinates differed in lane rows 16-31 / 48-63 whenever another stream's kernels shared
    // the CU — a correlation with the packed-fp32 code-generation switch (16 of 20 runs against 0 of 60), mechanism not isolated,
    // stand-alone reproducers negative (the library is built without packed fp32 instructions since).
    const float* hmv = homog + (long long)__builtin_amdgcn_readfirstlane(v) * 12;      // launch_fuse_points requires P % 8 == 0
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int dz = q * 8 + c4;
      cix[q] = ciy[q] = 0.f;
      if (dz < D) warp_coords_h(hmv, (float)x, (float)y, depths[b * D + dz], H, W, cix[q], ciy[q]);
    }
    for (int dz = 0; dz < D; ++dz) {
      const int q = dz >> 3, srcl = grp + (dz & 7);
      float ix = __shfl(q == 0 ? cix[0] : (q == 1 ? cix[1] : cix[2]), srcl);
      float iy = __shfl(q == 0 ? ciy[0] : (q == 1 ? ciy[1] : ciy[2]), srcl);
      const bool fin = isfinite(ix) && isfinite(iy);
      ix = fminf(fmaxf(fin ? ix : 0.f, -4.f), 1.0e6f);
      iy = fminf(fmaxf(fin ? iy : 0.f, -4.f), 1.0e6f);
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float tx = ix - fx, ty = iy - fy;
      const bool xin0 = (unsigned)x0 < (unsigned)W, xin1 = (unsigned)(x0 + 1) < (unsigned)W;
      const bool yin0 = (unsigned)y0 < (unsigned)H, yin1 = (unsigned)(y0 + 1) < (unsigned)H;
      const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
      const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
      float s00[4], s01[4], s10[4], s11[4];
      load4(src + ((long long)yc0 * W + xc0) * C, s00);
      load4(src + ((long long)yc0 * W + xc1) * C, s01);
      load4(src + ((long long)yc1 * W + xc0) * C, s10);
      load4(src + ((long long)yc1 * W + xc1) * C, s11);
      const float w00 = (xin0 && yin0) ? (1.f - tx) * (1.f - ty) : 0.f, w01 = (xin1 && yin0) ? tx * (1.f - ty) : 0.f;
      const float w10 = (xin0 && yin1) ? (1.f - tx) * ty : 0.f, w11 = (xin1 && yin1) ? tx * ty : 0.f;
      const float pr = prob[vp * D + dz];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // same accumulation order as the branchy version: corners 00, 01, 10, 11 (an absent corner adds nothing)
        float wv = 0.f;
        if (xin0 && yin0) wv += s00[e] * w00;
        if (xin1 && yin0) wv += s01[e] * w01;
        if (xin0 && yin1) wv += s10[e] * w10;
        if (xin1 && yin1) wv += s11[e] * w11;
        if (!fin) wv = __builtin_nanf("");
        float f = ref[e] + wv;
        if (ROUND_BF16) f = round_through<T>(f);           // the volume the cost net saw was stored in the 16-bit type
        acc[e] += f * pr;
      }
    }
    store4(out + vp * ldo + ch_off + c4 * 4, acc);
  }
}

int launch_fuse_points(int dtype, const void* feat, const float* homog, const float* depths, const int* choose,
                       const float* prob, float* out, int V, int B, int P, int D, int H, int W, int ldo, int ch_off,
                       hipStream_t s, int Vn) {
  if (Vn < 0) Vn = V;
  const long long total = (long long)Vn * P * 8;
  RGBM_REQUIRE(D >= 1 && D <= 24 && total > 0 && (total + 255) / 256 < (1ll << 31), "fuse_points supports up to 24 depth planes");
  RGBM_REQUIRE((P & 7) == 0, "fuse_points: the point count must be a multiple of 8 (a wave's 8 points share one view's homography)");
  const unsigned g = (unsigned)((total + 255) / 256);
  if (dtype == BF16)
    hipLaunchKernelGGL((fuse_points_kernel<unsigned short, true>), dim3(g), dim3(256), 0, s, (const unsigned short*)feat,
                       homog, depths, choose, prob, out, V, B, P, D, H, W, ldo, ch_off, Vn);
  else if (dtype == F16)
    hipLaunchKernelGGL((fuse_points_kernel<f16_t, true>), dim3(g), dim3(256), 0, s, (const f16_t*)feat,
                       homog, depths, choose, prob, out, V, B, P, D, H, W, ldo, ch_off, Vn);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL((fuse_points_kernel<bx3_t, false>), dim3(g), dim3(256), 0, s, (const bx3_t*)feat,
                       homog, depths, choose, prob, out, V, B, P, D, H, W, ldo, ch_off, Vn);
  else
    hipLaunchKernelGGL((fuse_points_kernel<float, false>), dim3(g), dim3(256), 0, s, (const float*)feat, homog, depths,
                       choose, prob, out, V, B, P, D, H, W, ldo, ch_off, Vn);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- mean over the P points of each view
// in [V*P][C] (fp32, or fp16 for the pose MLP of bf16 nets) -> out [V][C] fp32.  Blocks of 256 threads: C/E
// threads cover a row with 16-byte loads (E = 4 or 8 channels), 256/(C/E) rows are read side by side, every thread sums
// its channels over its share of the rows in fp32, and the row groups are added through LDS.  (One 2- or 4-byte element per
// thread and row ran at 1.5 TB/s.)
// Round 5: kMeanSplit workgroups per view (a slice of P / kMeanSplit points each, partial sums into `part`) and a second launch that adds
// the slices in a fixed order - one workgroup per view read its 256-512 KB through one CU's latency (20 + 38 us of a 1.49 ms forward at
// B = 1).  The split does not depend on the batch size, so neither does the order of the sums.
constexpr int kMeanSplit = 8;
template <typename TI>
__global__ __launch_bounds__(256) void mean_points_kernel(const TI* __restrict__ in, float* __restrict__ partial, int P, int C) {
  constexpr int E = 16 / sizeof(TI);
  __shared__ float part[256 * E];
  const int v = blockIdx.x / kMeanSplit, sl = blockIdx.x % kMeanSplit;
  const int p0 = (int)((long long)P * sl / kMeanSplit), p1 = (int)((long long)P * (sl + 1) / kMeanSplit);
  const int tpr = C / E, rows = 256 / tpr;                 // threads per row, rows in flight
  const int cc = threadIdx.x % tpr, rp = threadIdx.x / tpr;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  if (rp < rows)
#pragma unroll 8
    for (int p = p0 + rp; p < p1; p += rows) {               // unrolled: eight 16-byte loads in flight per thread
      float t[E];
      unpack_chunk(*reinterpret_cast<const uint4*>(in + ((long long)v * P + p) * C + cc * E), t, TI());
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] += t[e];
    }
  if (rp < rows) {
#pragma unroll
    for (int e = 0; e < E; ++e) part[rp * C + cc * E + e] = acc[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float sum = 0.f;
    for (int r = 0; r < rows; ++r) sum += part[r * C + c];
    partial[(long long)blockIdx.x * C + c] = sum;
  }
}

// the mean of channel c of view v from the slices' partial sums: the slices in ascending order, then one division — the one expression
// of mean_points_finish_kernel and of the consumers that finish the mean themselves (view_linear_kernel, pose_heads_kernel)
__device__ __forceinline__ float mean_from_partials(const float* __restrict__ partial, long long v, int c, int C, int P) {
  float sum = 0.f;
#pragma unroll
  for (int sl = 0; sl < kMeanSplit; ++sl) sum += partial[(v * kMeanSplit + sl) * C + c];
  return sum / (float)P;
}

__global__ __launch_bounds__(256) void mean_points_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int P, int C) {
  const int v = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) out[(long long)v * C + c] = mean_from_partials(partial, v, c, C, P);
}

// the slices' partial sums only (scratch: V * 8 * C floats): for a consumer that finishes the mean itself (launch_view_linear_mean,
// launch_pose_heads_mean: one launch less per mean, which is what a small-batch forward consists of)
int launch_mean_points_partial(int dtype, const void* in, float* scratch, int V, int P, int C, hipStream_t s) {
  RGBM_REQUIRE(dtype == F32 || dtype == F16 || dtype == BF16X3, "mean_points: fp32, fp16 or split-pair input");
  RGBM_REQUIRE(scratch != nullptr && (const void*)scratch != in, "mean_points: scratch");
  const int E = dtype == F16 ? 8 : 4;
  RGBM_REQUIRE(C % E == 0 && C / E <= 256 && 256 % (C / E) == 0, "mean_points: channel count");
  const dim3 g((unsigned)V * kMeanSplit);
  if (dtype == F16)
    hipLaunchKernelGGL(mean_points_kernel<f16_t>, g, dim3(256), 0, s, (const f16_t*)in, scratch, P, C);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(mean_points_kernel<bx3_t>, g, dim3(256), 0, s, (const bx3_t*)in, scratch, P, C);
  else
    hipLaunchKernelGGL(mean_points_kernel<float>, g, dim3(256), 0, s, (const float*)in, scratch, P, C);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// scratch: V * 8 * C floats (any buffer that is free between the producer of `in` and the consumer of `out`)
int launch_mean_points(int dtype, const void* in, float* out, float* scratch, int V, int P, int C, hipStream_t s) {
  RGBM_REQUIRE(dtype == F32 || dtype == F16 || dtype == BF16X3, "mean_points: fp32, fp16 or split-pair input");
  RGBM_REQUIRE(scratch != nullptr && (const void*)scratch != in && scratch != out, "mean_points: scratch");
  const int E = dtype == F16 ? 8 : 4;
  RGBM_REQUIRE(C % E == 0 && C / E <= 256 && 256 % (C / E) == 0, "mean_points: channel count");
  const dim3 g((unsigned)V * kMeanSplit);
  if (dtype == F16)
    hipLaunchKernelGGL(mean_points_kernel<f16_t>, g, dim3(256), 0, s, (const f16_t*)in, scratch, P, C);
  else if (dtype == BF16X3)
    hipLaunchKernelGGL(mean_points_kernel<bx3_t>, g, dim3(256), 0, s, (const bx3_t*)in, scratch, P, C);
  else
    hipLaunchKernelGGL(mean_points_kernel<float>, g, dim3(256), 0, s, (const float*)in, scratch, P, C);
  hipLaunchKernelGGL(mean_points_finish_kernel, dim3(V), dim3(256), 0, s, (const float*)scratch, out, P, C);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- small dense layer on per-view vectors
// out[v][o] = act(bias[o] + sum_i W[o][i0 + i] * x[v][i]),  W row-major [O][ldw].  One block per view.
// partial != null: x is not read — the input vector is the mean of launch_mean_points_partial's slices over P points, finished here
// and also written to mean_out [V][I] (the tensor launch_mean_points would have produced)
__global__ __launch_bounds__(256) void view_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ out, int I,
                                                          int O, int ldw, int i0, int relu, const float* __restrict__ partial, int P,
                                                          float* __restrict__ mean_out) {
  extern __shared__ float xs[];
  const int v = blockIdx.x;
  if (partial) {
    for (int i = threadIdx.x; i < I; i += blockDim.x) {
      const float m = mean_from_partials(partial, v, i, I, P);
      xs[i] = m;
      mean_out[(long long)v * I + i] = m;
    }
  } else {
    for (int i = threadIdx.x; i < I; i += blockDim.x) xs[i] = x[(long long)v * I + i];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < O; o += blockDim.x) {
    float acc = bias ? bias[o] : 0.f;
    const float* wr = W + (long long)o * ldw + i0;
    for (int i = 0; i < I; ++i) acc = fmaf(wr[i], xs[i], acc);
    if (relu) acc = acc < 0.f ? 0.f : acc;        // NaN propagates, like torch.relu
    out[(long long)v * O + o] = acc;
  }
}

int launch_view_linear(const float* x, const float* W, const float* bias, float* out, int V, int I, int O, int ldw, int i0,
                       int relu, hipStream_t s) {
  hipLaunchKernelGGL(view_linear_kernel, dim3(V), dim3(256), I * sizeof(float), s, x, W, bias, out, I, O, ldw, i0, relu, (const float*)nullptr, 0,
                     (float*)nullptr);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_view_linear_mean(const float* partial, int P, float* mean_out, const float* W, const float* bias, float* out, int V, int I, int O,
                            int ldw, int i0, int relu, hipStream_t s) {
  RGBM_REQUIRE(partial && mean_out && P > 0, "view_linear_mean arguments");
  hipLaunchKernelGGL(view_linear_kernel, dim3(V), dim3(256), I * sizeof(float), s, (const float*)nullptr, W, bias, out, I, O, ldw, i0, relu, partial, P,
                     mean_out);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- the three regression heads in one launch
// rotation / translation / size estimators (network_v5.py:485-494): Linear(256,256) ReLU Linear(256,128) ReLU Linear(128,6|3|3) on the
// pooled pose feature of a view.  One workgroup per (view, head); every output is the same sequential fma chain view_linear_kernel
// runs (bit-identical), the hidden vectors stay in LDS: one launch instead of nine (8 % of a B = 1 forward were launch latencies
// of these and of the staging copies below).
struct PoseHeadsDesc {
  const float* w[3][3];
  const float* b[3][3];
  float* out[3];
  int odim[3];
  const float* partial; int P; float* mean_out;      // partial != null: pf2 = the mean of mean_points' slices, finished here (head 0's workgroup writes it to mean_out)
  float* R;                                          // != null: head 0's six outputs also as a rotation matrix (ortho6d_rotation)
};

__device__ __forceinline__ void ortho6d_rotation(const float* a, float* o);

__global__ __launch_bounds__(256) void pose_heads_kernel(const float* __restrict__ pf2, const PoseHeadsDesc d) {
  __shared__ float x0[256], x1[256], x2[128], y6[8];
  const int v = blockIdx.x, h = blockIdx.y, t = threadIdx.x;
  if (d.partial) {
    const float m = mean_from_partials(d.partial, v, t, 256, d.P);
    x0[t] = m;
    if (h == 0) d.mean_out[(long long)v * 256 + t] = m;
  } else {
    x0[t] = pf2[(long long)v * 256 + t];
  }
  __syncthreads();
  {
    float acc = d.b[h][0][t];
    const float* wr = d.w[h][0] + (long long)t * 256;
    for (int i = 0; i < 256; ++i) acc = fmaf(wr[i], x0[i], acc);
    x1[t] = acc < 0.f ? 0.f : acc;               // NaN propagates, like torch.relu
  }
  __syncthreads();
  if (t < 128) {
    float acc = d.b[h][1][t];
    const float* wr = d.w[h][1] + (long long)t * 256;
    for (int i = 0; i < 256; ++i) acc = fmaf(wr[i], x1[i], acc);
    x2[t] = acc < 0.f ? 0.f : acc;
  }
  __syncthreads();
  if (t < d.odim[h]) {
    float acc = d.b[h][2][t];
    const float* wr = d.w[h][2] + (long long)t * 128;
    for (int i = 0; i < 128; ++i) acc = fmaf(wr[i], x2[i], acc);
    d.out[h][(long long)v * d.odim[h] + t] = acc;
    if (t < 8) y6[t] = acc;
  }
  if (h == 0 && d.R != nullptr) {      // block-uniform
    __syncthreads();
    if (t == 0) ortho6d_rotation(y6, d.R + (long long)v * 9);
  }
}

static int pose_heads_launch(const float* pf2, const float* partial, int P, float* mean_out, float* R, float* const w[3][3],
                             float* const b[3][3], float* const out[3], const int odim[3], int V, hipStream_t s) {
  PoseHeadsDesc d;
  for (int h = 0; h < 3; ++h) {
    for (int l = 0; l < 3; ++l) { d.w[h][l] = w[h][l]; d.b[h][l] = b[h][l]; }
    d.out[h] = out[h];
    d.odim[h] = odim[h];
    RGBM_REQUIRE(odim[h] <= 8, "pose heads: at most 8 outputs per head");
  }
  RGBM_REQUIRE(R == nullptr || odim[0] == 6, "pose heads: the rotation matrix needs head 0's six outputs");
  d.partial = partial; d.P = P; d.mean_out = mean_out; d.R = R;
  hipLaunchKernelGGL(pose_heads_kernel, dim3(V, 3), dim3(256), 0, s, pf2, d);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_pose_heads(const float* pf2, float* const w[3][3], float* const b[3][3], float* const out[3], const int odim[3], int V,
                      hipStream_t s) {
  return pose_heads_launch(pf2, nullptr, 0, nullptr, nullptr, w, b, out, odim, V, s);
}

// the heads on the mean of launch_mean_points_partial's slices (finished here, written to mean_out [V][256]); R != null: head 0's
// six outputs also as rotation matrices [V][9] (launch_ortho6d's arithmetic) — three launches of a small-batch forward in one
int launch_pose_heads_mean(const float* partial, int P, float* mean_out, float* R, float* const w[3][3], float* const b[3][3],
                           float* const out[3], const int odim[3], int V, hipStream_t s) {
  RGBM_REQUIRE(partial && mean_out && P > 0, "pose_heads_mean arguments");
  return pose_heads_launch(nullptr, partial, P, mean_out, R, w, b, out, odim, V, s);
}

// ---------------------------------------------------------------- input / output staging of a forward in one launch each
// (were 4 + 10 device-to-device copies of a few KB: per-launch latency is what a small-batch forward consists of)
__global__ void stage_in_kernel(const float* __restrict__ P1, const float* __restrict__ P2, const int* __restrict__ c1,
                                const int* __restrict__ c2, float* __restrict__ Pviews, int* __restrict__ choose, int B, int P) {
  const long long nP = (long long)B * 16, nC = (long long)B * P;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < 2 * (nP + nC); i += (long long)gridDim.x * blockDim.x) {
    if (i < nP) Pviews[i] = P1[i];
    else if (i < 2 * nP) Pviews[i] = P2[i - nP];
    else if (i < 2 * nP + nC) choose[i - 2 * nP] = c1[i - 2 * nP];
    else choose[i - 2 * nP] = c2[i - 2 * nP - nC];
  }
}

int launch_stage_in(const float* P1, const float* P2, const int* c1, const int* c2, float* Pviews, int* choose, int B, int P, hipStream_t s) {
  const long long n = 2ll * B * (16 + P);
  hipLaunchKernelGGL(stage_in_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, P1, P2, c1, c2, Pviews, choose, B, P);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

struct StageOutDesc {
  const float *nocs4, *depth, *R, *tv, *sv;      // [V*P][4], [V*P], [V][9], [V][3], [V][3]; views = [view-1 batch ; view-2 batch]
  float *nocs[2], *dep[2], *r[2], *t[2], *s[2];  // reference shapes: [B,P,3], [B,P], [B,3,3], [B,3], [B,3]
  int B, P, view2;                               // view2 = 0: the view-2 outputs were not computed and are filled with NaN
};

__global__ void stage_out_kernel(const StageOutDesc d) {
  const long long BP = (long long)d.B * d.P;
  const long long n_nocs = BP * 3, n_small = (long long)d.B * 15, per_side = n_nocs + BP + n_small;
  const float qnan = __uint_as_float(0xffffffffu);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < 2 * per_side; i += (long long)gridDim.x * blockDim.x) {
    const int side = i >= per_side;
    long long j = i - side * per_side;
    const bool have = side == 0 || d.view2;
    if (j < n_nocs) {
      const long long pt = j / 3;
      const int c = (int)(j - pt * 3);
      d.nocs[side][j] = have ? d.nocs4[(side * BP + pt) * 4 + c] : qnan;
    } else if ((j -= n_nocs) < BP) {
      d.dep[side][j] = have ? d.depth[side * BP + j] : qnan;
    } else {
      j -= BP;
      const long long b = j / 15;
      const int e = (int)(j - b * 15);
      const long long v = (long long)side * d.B + b;
      if (e < 9) d.r[side][b * 9 + e] = have ? d.R[v * 9 + e] : qnan;
      else if (e < 12) d.t[side][b * 3 + (e - 9)] = have ? d.tv[v * 3 + (e - 9)] : qnan;
      else d.s[side][b * 3 + (e - 12)] = have ? d.sv[v * 3 + (e - 12)] : qnan;
    }
  }
}

int launch_stage_out(const float* nocs4, const float* depth, const float* R, const float* tv, const float* sv, float* const nocs[2],
                     float* const dep[2], float* const r[2], float* const t[2], float* const sz[2], int B, int P, int view2, hipStream_t s) {
  StageOutDesc d;
  d.nocs4 = nocs4; d.depth = depth; d.R = R; d.tv = tv; d.sv = sv;
  for (int k = 0; k < 2; ++k) { d.nocs[k] = nocs[k]; d.dep[k] = dep[k]; d.r[k] = r[k]; d.t[k] = t[k]; d.s[k] = sz[k]; }
  d.B = B; d.P = P; d.view2 = view2;
  const long long n = 2ll * ((long long)B * P * 4 + (long long)B * 15);
  hipLaunchKernelGGL(stage_out_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, d);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- Ortho6d -> rotation matrix (rotation_utils.py:18-27)
__device__ __forceinline__ void ortho6d_rotation(const float* a, float* o) {      // a: [x_raw(3), y_raw(3)]; o: 3 x 3, columns [x y z]
  float xr[3] = {a[0], a[1], a[2]}, y[3] = {a[3], a[4], a[5]}, z[3], x[3];
  float n = fmaxf(sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]), 1e-8f);
  for (int i = 0; i < 3; ++i) y[i] /= n;
  z[0] = xr[1] * y[2] - xr[2] * y[1];
  z[1] = xr[2] * y[0] - xr[0] * y[2];
  z[2] = xr[0] * y[1] - xr[1] * y[0];
  n = fmaxf(sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]), 1e-8f);
  for (int i = 0; i < 3; ++i) z[i] /= n;
  x[0] = y[1] * z[2] - y[2] * z[1];
  x[1] = y[2] * z[0] - y[0] * z[2];
  x[2] = y[0] * z[1] - y[1] * z[0];
  for (int i = 0; i < 3; ++i) { o[i * 3 + 0] = x[i]; o[i * 3 + 1] = y[i]; o[i * 3 + 2] = z[i]; }
}

__global__ void ortho6d_kernel(const float* __restrict__ r6, float* __restrict__ R, int V) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  ortho6d_rotation(r6 + (long long)v * 6, R + (long long)v * 9);
}

int launch_ortho6d(const float* r6, float* R, int V, hipStream_t s) {
  hipLaunchKernelGGL(ortho6d_kernel, dim3((V + 63) / 64), dim3(64), 0, s, r6, R, V);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- strided row copy (fp32), e.g. [rows][4] -> [rows][3]
__global__ void copy_cols_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows, int ldi, int ldo, int n) {
  const long long total = rows * n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / n;
    const int c = (int)(i - r * n);
    out[r * ldo + c] = in[r * ldi + c];
  }
}

int launch_copy_cols(const float* in, float* out, long long rows, int ldi, int ldo, int n, hipStream_t s) {
  const long long total = rows * n;
  unsigned g = (unsigned)((total + 255) / 256);
  if (g > 65536) g = 65536;
  if (g == 0) g = 1;
  hipLaunchKernelGGL(copy_cols_kernel, dim3(g), dim3(256), 0, s, in, out, rows, ldi, ldo, n);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
