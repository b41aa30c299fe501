// Does v_mfma_f32_16x16x32_f16 on gfx950 keep fp16 denormal inputs?  (tools/split_emulation.py: an fp16 hi + lo split only
// reaches 3e-6 if it does; with denormals flushed it is no better than plain fp16.)   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float* out, float av, float bv) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  a[0] = (_Float16)av; b[0] = (_Float16)bv;          // only k = 0 of each lane group contributes
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  const float tests[3][2] = {{1.f, 1.f}, {9.5367431640625e-07f /* 2^-20, fp16 denormal */, 1.f}, {3.0517578125e-05f /* 2^-15 denormal */, 4.f}};
  for (auto& t : tests) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
    float h = -1.f; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a=%g b=%g  -> mfma sum = %.9g (expected %.9g x4 lane groups = %.9g)\n", t[0], t[1], h, t[0] * t[1], 4 * t[0] * t[1]);
  }
  return 0;
}
