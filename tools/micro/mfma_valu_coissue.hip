// Micro-benchmark: how much VALU work of ONE wave fits beside the MFMA stream of ANOTHER wave on the same SIMD?
// One workgroup per CU, 8 waves: waves 0-3 (one per SIMD) issue NM independent v_mfma_f32_16x16x32_bf16 per iteration (six accumulators
// in rotation, like the plane sweep's consumers), waves 4-7 issue NV VALU instructions of one kind per iteration.  Reports cycles per
// iteration (s_memtime) of each role alone and of both together.  Round 5: the plane sweep's time per plane equals the SUM of its two
// roles' times alone - this asks whether that is a property of the SIMD or of that kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_valu_coissue tools/micro/mfma_valu_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// KIND: 0 v_fma_f32, 1 v_and + v_lshlrev (unpack pair), 2 v_cvt_pk_bf16_f32, 3 v_dot2_f32_bf16, 4 v_perm_b32, 5 v_pk_fma_f32, 6 v_fma_mix_f32, 7 v_pk_fma_f16
template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, int do_mfma, int do_valu, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (do_mfma) {
      f32x4 c[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      unsigned a4[4] = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f00u, 0x3f803f80u};
      unsigned b4[4] = {0x3f803f80u, 0x3f003f00u + lane, 0x3f803f80u, 0x3f803f80u};
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      u4 a = {a4[0], a4[1], a4[2], a4[3]}, b = {b4[0], b4[1], b4[2], b4[3]};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 54; ++m)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[m % 6]) : "v"(a), "v"(b));
      }
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]));
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) s += c[i][0];
      if (s == 1234.5f) sink[threadIdx.x] = s;
    }
  } else {
    if (do_valu) {
      float v[8]; unsigned u[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { v[i] = 1.f + lane * 1e-3f + i; u[i] = 0x3f803f80u + lane + i; }
      const float w = 0.999f; const unsigned wp = 0x3f7f3f7fu;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 40; ++m) {          // 40 x 8 = 320 VALU per iteration, eight independent chains
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(w));
            else if (KIND == 1) { if (m & 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i])); else asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i])); }
            else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(v[i]));
            else if (KIND == 3) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(u[i]), "v"(wp));
            else if (KIND == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(wp), "v"(0x05040100u));
            else if (KIND == 5) { f32x2 p = {v[i], v[i]}; asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p) : "v"(p)); v[i] = p.x; }
            else if (KIND == 6) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(u[i]), "v"(w));
            else if (KIND == 7) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(u[i]) : "v"(wp));
          }
        }
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i] + __uint_as_float(u[i]);
      if (s == 1234.5f) sink[threadIdx.x] = s;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int KIND>
static void run(const char* name, unsigned long long* out, float* sink) {
  const int iters = 2000;
  double res[3][2];
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int dm = cfg != 1, dv = cfg != 0;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, iters / 10, dm, dv, out, sink);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, iters, dm, dv, out, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part: convert with the measured wall clock instead -> report ratios + raw ticks
    res[cfg][0] = (double)h[0] / iters; res[cfg][1] = (double)h[4] / iters;
  }
  printf("%-28s ticks/iter: MFMA alone %.2f | VALU alone %.2f | together: MFMA wave %.2f, VALU wave %.2f  -> together / (sum of alone) = %.2f, / max = %.2f\n", name,
         res[0][0], res[1][1], res[2][0], res[2][1], (res[2][0] > res[2][1] ? res[2][0] : res[2][1]) / (res[0][0] + res[1][1]),
         (res[2][0] > res[2][1] ? res[2][0] : res[2][1]) / (res[0][0] > res[1][1] ? res[0][0] : res[1][1]));
}

int main() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 4096);
  printf("per iteration: 54 MFMA 16x16x32 bf16 (864 matrix-pipe cycles) on waves 0-3, 320 VALU of one kind on waves 4-7 (same SIMDs)\n");
  run<0>("v_fma_f32", out, sink);
  run<1>("v_and / v_lshlrev", out, sink);
  run<2>("v_cvt_pk_bf16_f32", out, sink);
  run<3>("v_dot2_f32_bf16", out, sink);
  run<4>("v_perm_b32", out, sink);
  run<5>("v_pk_fma_f32", out, sink);
  run<6>("v_fma_mix_f32", out, sink);
  run<7>("v_pk_fma_f16", out, sink);
  return 0;
}
