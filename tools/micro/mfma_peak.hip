// Micro-benchmark: how fast does one CU retire v_mfma_f32_16x16x32_bf16 in the shape of the implicit-GEMM multiply loop?
//   mode 0: MFMAs only (32 per iteration, 16 accumulators), mode 1: + 16 ds_read_b128 per iteration (fragments from LDS),
//   mode 2: as 1 with a workgroup barrier per iteration.  blockDim = 64 * waves; one workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/micro/mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
static int g_iters = 20000;          // argv[1]: iterations per timed launch (200000 = ~0.1 s per line: shows clock throttling)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// data pattern: 0 = every element 1.0 (few bits toggle: the matrix pipe draws little power), 1 = pseudo-random bf16 in (-2, 2)
__device__ int g_random = 0;
__device__ __forceinline__ uint4 fill_value(int i) {
  if (!g_random) return make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  unsigned h = (unsigned)i * 2654435761u;
  uint4 v;
  unsigned* p = &v.x;
  for (int k = 0; k < 4; ++k) {
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
    // two bf16 with exponent 0x3f (values in [1,2)) or 0x3e, random sign and mantissa
    p[k] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
  }
  return v;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  __shared__ uint4 lds[6144];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 6144; i += blockDim.x) lds[i] = fill_value(i);
  __syncthreads();
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = lds[lane + 64 * i]; b[i] = lds[lane + 64 * (4 + i)]; }
  const uint4* base = lds + (tid >> 6) * 512 + lane;
  for (int it = 0; it < iters; ++it) {
    for (int s = 0; s < 2; ++s) {
      if (MODE >= 1) {
        for (int i = 0; i < 4; ++i) { a[i] = base[(s * 8 + i) * 64 % 5632]; b[i] = base[(s * 8 + 4 + i) * 64 % 5632]; }
      }
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i * 4 + j], 0, 0, 0);
    }
    if (MODE == 2) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + tid] = s;
}

// mode 3: the multiply loop as hand-ordered: 8 fragment reads of the NEXT half issued from inline asm (invisible to hipcc's
// wait-count insertion), ONE s_waitcnt for the current half, then 16 MFMAs back to back.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ void lds_read16(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
__global__ __launch_bounds__(768) void k3(float* out, int iters, int barrier) {
  __shared__ uint4 lds[6144];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 6144; i += blockDim.x) lds[i] = fill_value(i);
  __syncthreads();
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 f0[8], f1[8];
  const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds + ((tid >> 6) * 512 + lane) * 16;
  for (int i = 0; i < 8; ++i) lds_read16(f0[i], base + i * 1024);
  auto mma = [&](u32x4 (&f)[8]) {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f[i]), __builtin_bit_cast(bf16x8, f[4 + j]), acc[i * 4 + j], 0, 0, 0);
  };
#define TOUCH8(F) "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7])
  for (int it = 0; it < iters; ++it) {
    for (int i = 0; i < 8; ++i) lds_read16(f1[i], base + (8 + i) * 1024);
    asm volatile("s_waitcnt lgkmcnt(8)" : TOUCH8(f0) :: "memory");
    mma(f0);
    for (int i = 0; i < 8; ++i) lds_read16(f0[i], base + i * 1024);
    asm volatile("s_waitcnt lgkmcnt(8)" : TOUCH8(f1) :: "memory");
    mma(f1);
    if (barrier) { asm volatile("s_waitcnt lgkmcnt(0)" : TOUCH8(f0) :: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : TOUCH8(f0) :: "memory");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + tid] = s + (float)f0[0][0];
}

void run3(int waves, int barrier, const char* name) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = g_iters;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k3, dim3(256), dim3(64 * waves), 0, 0, out, 100, barrier);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k3, dim3(256), dim3(64 * waves), 0, 0, out, iters, barrier);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * waves * iters * 32.0 * 16 * 16 * 32 * 2;
  printf("%-34s waves/CU %2d: %.3f ms  %.0f TFLOP/s  (%.1f clk per MFMA per SIMD at 2.4 GHz)\n", name, waves, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (iters * 32.0 * waves / 4.0));
  hipFree(out);
}


// mode 4: register tile FM x FN per wave (FM + FN fragment reads per FM * FN MFMAs and half K tile), plain loads, compiler-scheduled with the
// ws kernel's half-tile software pipeline (reads of the next half issued before the MFMAs of the current one), barrier per K tile.
// 8 waves of 4 x 4 = the ws kernel's multiply loop; 4 waves of 8 x 4 = one multiply wave per SIMD with a 128 x 64 tile.
template <int FM, int FN>
__global__ __launch_bounds__(512) void k4(float* out, int iters, int barrier) {
  __shared__ uint4 lds[6144];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 6144; i += blockDim.x) lds[i] = fill_value(i);
  __syncthreads();
  f32x4 acc[FM][FN];
  for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 a0[FM], b0[FN], a1[FM], b1[FN];
  const uint4* base = lds + (tid >> 6) * 256 + lane;
  auto load = [&](int s, uint4 (&a)[FM], uint4 (&b)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i) a[i] = base[((s * (FM + FN) + i) * 64) % 4096];
#pragma unroll
    for (int j = 0; j < FN; ++j) b[j] = base[((s * (FM + FN) + FM + j) * 64) % 4096];
  };
  auto mma = [&](const uint4 (&a)[FM], const uint4 (&b)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
  };
  load(1, a1, b1);
  for (int it = 0; it < iters; ++it) {
    if (barrier) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    load(0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    load(1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + tid] = s;
}
template <int FM, int FN>
void run4(int waves, int barrier, const char* name) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = g_iters;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k4<FM, FN>), dim3(256), dim3(64 * waves), 0, 0, out, 100, barrier);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k4<FM, FN>), dim3(256), dim3(64 * waves), 0, 0, out, iters, barrier);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double nm = 2.0 * FM * FN;
  const double flops = 256.0 * waves * iters * nm * 16 * 16 * 32 * 2;
  printf("%-34s waves/CU %2d: %.3f ms  %.0f TFLOP/s  (%.1f clk per MFMA per SIMD at 2.4 GHz)\n", name, waves, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (iters * nm * waves / 4.0));
  hipFree(out);
}

template <int MODE>
void run(int waves, const char* name) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = g_iters;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * waves * iters * 32.0 * 16 * 16 * 32 * 2;
  printf("%-34s waves/CU %2d: %.3f ms  %.0f TFLOP/s  (%.1f clk per MFMA per SIMD at 2.4 GHz)\n", name, waves, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (iters * 32.0 * waves / 4.0));
  hipFree(out);
}

int main(int argc, char** argv) {
  if (argc > 1) g_iters = atoi(argv[1]);
  for (int rnd = 0; rnd < 2; ++rnd) {
  hipMemcpyToSymbol(HIP_SYMBOL(g_random), &rnd, sizeof(int));
  printf("---- operand data: %s ----\n", rnd ? "pseudo-random bf16" : "all 1.0");
  for (int w : {4, 8, 12}) run<0>(w, "MFMA only");
  for (int w : {4, 8, 12}) run<1>(w, "MFMA + 16 ds_read_b128 / 32");
  for (int w : {4, 8, 12}) run<2>(w, "+ barrier per 32 MFMAs");
  for (int w : {4, 8, 12}) run3(w, 0, "asm reads, 1 wait per 16 MFMAs");
  for (int w : {4, 8, 12}) run3(w, 1, "asm reads, 1 wait, + barrier");
  run4<4, 4>(8, 1, "4x4 tile, pipelined, + barrier");
  run4<4, 4>(8, 0, "4x4 tile, pipelined, no barrier");
  run4<8, 4>(4, 1, "8x4 tile, pipelined, + barrier");
  run4<8, 4>(4, 0, "8x4 tile, pipelined, no barrier");
  run4<8, 4>(8, 1, "8x4 tile, 8 waves, + barrier");
  run4<8, 8>(4, 1, "8x8 tile, pipelined, + barrier");
  }
  return 0;
}
