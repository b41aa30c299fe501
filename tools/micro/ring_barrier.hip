// The plane ring of the sweep kernels in isolation: 4 producer waves write "plane z" (every dword = z) into slot z % NS of an LDS
// ring, `s_waitcnt lgkmcnt(0)`, `s_barrier`; 3 consumer waves read plane z - 1 from slot (z - 1) % NS during iteration z
// (36 ds_read_b128 spread over a burst of MFMAs), `s_waitcnt lgkmcnt(0)`, `s_barrier`.  By the protocol NS = 2 is enough: a
// slot is rewritten in iteration z + 1, after the barrier that closes the iteration in which it was read.  Any dword that is
// not z - 1 is a violation.  Variants: NS = 2 / 3; consumer slower or faster than the producers (extra MFMAs / extra VALU).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/ring_barrier.hip -o tools/micro/ring_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
constexpr int NV = 252, VS = 80, SLOT = NV * VS;

template <int NS, int CONS_MFMA, int PROD_VALU>
__global__ __launch_bounds__(448, 3) void k(unsigned* out, int planes, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned bad = 0, first = 0xffffffffu;
  for (int t = 0; t < tiles; ++t) {
    if (wave < 4) {
      const bool act = tid < NV;
      float x = (float)tid;
      for (int z = 0; z <= planes; ++z) {
        if (act && z < planes) {
#pragma unroll
          for (int j = 0; j < PROD_VALU; ++j) x = x * 1.0001f + 0.5f;
          const unsigned v = (unsigned)(t * 1000 + z) + (x == 12345.f ? 1u : 0u);
          unsigned char* dst = ring + (z % NS) * SLOT + tid * VS;
          const u4 w = {v, v, v, v};
#pragma unroll
          for (int c = 0; c < 4; ++c) *reinterpret_cast<u4*>(dst + c * 16) = w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    } else {
      const int cw = wave - 4, lr = lane & 15, lg = lane >> 4;
      const int boff = ((cw * 4) * 18 + lr) * VS + lg * 16;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const u4 A = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      for (int z = 0; z <= planes; ++z) {
        if (z >= 1) {
          const int p = z - 1;
          const unsigned want = (unsigned)(t * 1000 + p);
          const unsigned char* slot = ring + (p % NS) * SLOT + boff;
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              const u4 b = *reinterpret_cast<const u4*>(slot + ((f + tp / 3) * 18 + tp % 3) * VS);
              if (b.x != want || b.y != want || b.z != want || b.w != want) { ++bad; if (first == 0xffffffffu) first = (unsigned)(p * 16 + tp); }
#pragma unroll
              for (int m = 0; m < CONS_MFMA; ++m) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc) : "v"(A));
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
      if (acc.x == 12345.f) bad += 1000000;
    }
  }
  if (wave >= 4) { atomicAdd(out, bad); atomicMin(out + 1, first); }
}

template <int NS, int CM, int PV>
void run(unsigned* d, const char* what) {
  unsigned init[2] = {0u, 0xffffffffu};
  hipMemcpy(d, init, 8, hipMemcpyHostToDevice);
  auto kern = k<NS, CM, PV>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NS * SLOT);
  hipLaunchKernelGGL(kern, dim3(4096), dim3(448), NS * SLOT, 0, d, 24, 8);
  unsigned h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("ring of %d slots, %-44s violations %u of %.0f reads%s\n", NS, what, h[0], 4096.0 * 8 * 24 * 36 * 3 * 64, h[0] ? "" : "");
}

int main() {
  unsigned* d; hipMalloc(&d, 8);
  run<2, 0, 0>(d, "consumer: reads only, producer: stores only:");
  run<2, 2, 0>(d, "consumer slow (2 MFMAs per read):");
  run<2, 0, 64>(d, "producer slow (64 VALU per plane):");
  run<2, 2, 230>(d, "both busy (kernel-like):");
  run<3, 2, 230>(d, "both busy (kernel-like):");
  run<3, 0, 0>(d, "reads / stores only:");
  return 0;
}
