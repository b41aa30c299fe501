// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 (4 passes = 16 cycles nominal) as a function of how its operands change between
// consecutive instructions - one wave per SIMD, nothing else running.  Round 5: the plane sweep's consumers issue 54 of them per plane in
// 1080 cycles (20 each) with NO memory instruction in the loop, while the same count with constant A / B operands takes 872.
//   mode 0: A, B constant, six accumulators in rotation (in place)
//   mode 1: the sweep's order - 9 taps x 3 fragments x {A01[tap] -> Xn[f], A2[tap] -> Xp[f]}: 18 distinct A quads, 15 distinct B quads
//   mode 2: as 1 with v_mfma_f32_32x32x16_bf16 (8 passes): 27 instructions for the same executed flops
//   mode 3: as 1, but each B is used by four consecutive MFMAs (two fragments' worth of accumulators: 12 in rotation)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_operand_rate tools/micro/mfma_operand_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  u4 A[18], B[15];
#pragma unroll
  for (int i = 0; i < 18; ++i) A[i] = u4{0x3c003c00u + lane + i, 0x3c003c00u, 0x38003800u + i, 0x3c003c00u};
#pragma unroll
  for (int i = 0; i < 15; ++i) B[i] = u4{0x3c003c00u + i, 0x38003800u + lane, 0x3c003c00u, 0x3c003c00u + i};
  f32x4 X[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) X[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 Y[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) Y[i][e] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 18; ++i) asm volatile("" : "+v"(A[i]));
#pragma unroll
    for (int i = 0; i < 15; ++i) asm volatile("" : "+v"(B[i]));
    if (MODE == 0) {
#pragma unroll
      for (int m = 0; m < 54; ++m) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(X[m % 6]) : "v"(A[0]), "v"(B[0]));
    } else if (MODE == 1) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(X[f]) : "v"(A[tp]), "v"(B[(f + tp / 3) * 3 + tp % 3]));
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(X[3 + f]) : "v"(A[9 + tp]), "v"(B[(f + tp / 3) * 3 + tp % 3]));
        }
    } else if (MODE == 2) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int f = 0; f < 3; ++f) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(Y[f]) : "v"(A[tp + 9 * (f & 1)]), "v"(B[(f + tp / 3) * 3 + tp % 3]));
    } else {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          // same A for consecutive pairs, B changes: (A01, b0) (A01, b1) (A2, b0) (A2, b1) ordering over fragment pairs is not possible with 3
          // fragments; here: A01 for all three fragments, then A2 for all three
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(X[f]) : "v"(A[tp]), "v"(B[(f + tp / 3) * 3 + tp % 3]));
        }
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int f = 0; f < 3; ++f) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(X[3 + f]) : "v"(A[9 + tp]), "v"(B[(f + tp / 3) * 3 + tp % 3]));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += X[i][0];
#pragma unroll
  for (int i = 0; i < 3; ++i) s += Y[i][0];
  if (s == 1234.5f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE> static void run(const char* name, unsigned long long* out, float* sink) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, iters / 10, out, sink);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, iters, out, sink);
  (void)hipDeviceSynchronize();
  unsigned long long h = 0;
  (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("%-90s %7.1f cycles per iteration\n", name, (double)h / iters);
}

int main() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 4096);
  run<0>("54 x 16x16x32, constant A and B, 6 accumulators (nominal 864)", out, sink);
  run<1>("54 x 16x16x32 in the plane sweep's order (18 A quads, 15 B quads, 6 accumulators)", out, sink);
  run<3>("54 x 16x16x32, A constant over three consecutive instructions", out, sink);
  run<2>("27 x 32x32x16 over the same operands (nominal 864)", out, sink);
  return 0;
}
