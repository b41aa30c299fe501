// Does a wave's plain fp32 arithmetic change when waves of ANOTHER queue's kernel share its SIMD?  (Round 4, DESIGN 5d: two forwards
// overlapping on the device differ in fuse_points_kernel's warp coordinates although the registers that feed them are bit-identical.)
// Kernel A evaluates a warp-coordinate-like chain (FMAs, two divisions, floor) REP times per thread from inputs that depend only on
// the thread index, and counts evaluations that differ from the thread's first one.  Kernel B is a co-tenant on another stream:
//   mode 0 none, 1 an MFMA stream (4 waves, no LDS), 2 a VALU-only stream, 3 an LDS + MFMA kernel (halo-tile-conv-like footprint).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_under_mfma valu_under_mfma.hip ; run: ./valu_under_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float chain(float x, float y, float d, const float* hm) {
  const float rx = hm[0] * x + hm[1] * y + hm[2], ry = hm[3] * x + hm[4] * y + hm[5], rz = hm[6] * x + hm[7] * y + hm[8];
  const float px = rx * d + hm[9], py = ry * d + hm[10], pz = rz * d + hm[11];
  const float u = px / pz, v = py / pz;
  const float gx = u / 111.5f - 1.f, gy = v / 111.5f - 1.f;
  const float ix = ((gx + 1.f) * 224.f - 1.f) / 2.f, iy = ((gy + 1.f) * 224.f - 1.f) / 2.f;
  return floorf(ix) * 3.f + (ix - floorf(ix)) + iy;
}

__global__ void kernel_a(const float* __restrict__ hm_all, unsigned* __restrict__ bad, int reps) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float hm[12];
  for (int e = 0; e < 12; ++e) hm[e] = hm_all[(t % 251) * 12 + e];
  const float x = (float)(t % 224), y = (float)((t / 224) % 224);
  float first = 0.f;
  unsigned nbad = 0;
  for (int r = 0; r < reps; ++r) {
    float acc = 0.f;
    for (int dz = 0; dz < 24; ++dz) {
      float d = 0.1f + 0.1f * (float)dz;
      asm volatile("" : "+v"(d));                       // a fresh evaluation every time
      acc += chain(x, y, d, hm);
    }
    if (r == 0) first = acc; else nbad += __float_as_uint(acc) != __float_as_uint(first);
  }
  if (nbad) atomicAdd(bad + (threadIdx.x & 63), nbad);
}

__global__ __launch_bounds__(256) void kernel_mfma(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 a = make_uint4(0x3f803f80u + threadIdx.x, 0x3f813f80u, 0x3f823f80u, 0x3f833f80u), b = make_uint4(0x3f803f81u, 0x3f803f82u, 0x3f803f83u + threadIdx.x, 0x3f803f84u);
  for (int it = 0; it < iters; ++it)
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void kernel_valu(float* out, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0001f;
  for (int it = 0; it < iters * 32; ++it) { a = a * b + 0.5f; b = b * 0.99999f + 1e-5f; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}

__global__ __launch_bounds__(256) void kernel_lds_mfma(float* out, int iters) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < 1600; i += 256) lds[i] = make_uint4(0x3f803f80u + i, 0x3f813f80u, 0x3f823f80u, 0x3f833f80u);
  __syncthreads();
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const uint4 a = lds[(threadIdx.x + it * 7) % 1600], b = lds[(threadIdx.x * 3 + it) % 1600];
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    if ((it & 15) == 15) __syncthreads();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

int main() {
  float* hm; unsigned* bad; float* sink;
  hipMalloc(&hm, 251 * 12 * 4); hipMalloc(&bad, 64 * 4); hipMalloc(&sink, 4096 * 256 * 4);
  float h[251 * 12];
  for (int i = 0; i < 251; ++i) { float* p = h + i * 12; p[0] = 1.f + 0.001f * i; p[1] = 0.01f; p[2] = 3.f; p[3] = -0.02f; p[4] = 0.98f; p[5] = 2.f; p[6] = 1e-4f; p[7] = -2e-4f; p[8] = 1.f; p[9] = 20.f + i * 0.1f; p[10] = -10.f; p[11] = 0.05f; }
  hipMemcpy(hm, h, sizeof(h), hipMemcpyHostToDevice);
  hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
  const char* names[4] = {"kernel A alone", "kernel A next to an MFMA stream (other stream)", "kernel A next to a VALU stream (other stream)", "kernel A next to an LDS + MFMA kernel (other stream)"};
  for (int mode = 0; mode < 4; ++mode) {
    unsigned total = 0, rows[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 10; ++rep) {
      hipMemset(bad, 0, 64 * 4);
      hipDeviceSynchronize();
      if (mode == 1) hipLaunchKernelGGL(kernel_mfma, dim3(2048), dim3(256), 0, sb, sink, 40000);
      if (mode == 2) hipLaunchKernelGGL(kernel_valu, dim3(2048), dim3(256), 0, sb, sink, 40000);
      if (mode == 3) hipLaunchKernelGGL(kernel_lds_mfma, dim3(4096), dim3(256), 25600, sb, sink, 20000);
      for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(kernel_a, dim3(8192), dim3(256), 0, sa, hm, bad, 8);
      hipDeviceSynchronize();
      unsigned hb[64];
      hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l) { total += hb[l]; rows[l / 16] += hb[l]; }
    }
    printf("%-55s: %u differing evaluations (lane rows 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u)\n", names[mode], total, rows[0], rows[1], rows[2], rows[3]);
  }
  return 0;
}
