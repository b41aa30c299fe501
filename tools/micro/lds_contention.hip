// Micro-benchmark: does LDS-DMA traffic (global_load / buffer_load ... lds) slow down ds_read_b128 streams of other waves of the CU, and
// by how much?  One workgroup per CU: R reader waves stream ds_read_b128 over a 64 KB region (conflict-free, 16 reads in flight), D
// loader waves stream 1 KiB LDS-DMA pieces from an L2-resident buffer into another 64 KB region (12 in flight each).
// Reports bytes per clock (at the measured wall time and a nominal 2.4 GHz) of each side alone and together.
// Round 5: the implicit GEMM's K loop loses a quarter of its time only when BOTH the fragment reads and the LDS-DMA are present
// (each alone is free) - this measures the interference in isolation.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_contention tools/micro/lds_contention.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void blds16(unsigned voff, const __amdgpu_buffer_rsrc_t& rsrc, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds_off) : "memory");
}

// waves [0, R): readers; waves [R, R + D): loaders.  iters: per-wave loop count (readers: 16 reads each; loaders: 12 pieces each)
__global__ __launch_bounds__(1024) void k(const char* src, int R, int D, int it_r, int it_d, unsigned* out, int mfma) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds;
  if (wave < R) {
    unsigned acc = 0;
    unsigned addr = lbase + wave * 1024 + lane * 16;
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
    f32x16 c = {};
    for (int it = 0; it < it_r; ++it) {
      u32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(addr) : "memory");
        addr = lbase + ((addr - lbase + 8192) & 65535);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
      if (mfma) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v[i]), __builtin_bit_cast(bf16x8, v[(i + 1) & 7]), c, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= v[i][0];
      }
    }
    if (acc == 0x12345678u || c[0] == 1234.5f) out[blockIdx.x] = 1;
  } else if (wave < R + D) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)0x7fffffff, 0x00020000);
    const int pw = wave - R;
    unsigned row = ((unsigned)(blockIdx.x >> 3) * 40u + pw * 8u) & 2047u;
    const int j = lane & 7, r8 = lane >> 3;
    for (int it = 0; it < it_d; ++it) {
#pragma unroll
      for (int p = 0; p < 12; ++p) {
        const unsigned ao = ((row + r8) & 2047u) * 1024u + j * 16;
        row = (row + 8 * D) & 2047u;
        blds16(ao, rsrc, lbase + 65536 + ((pw * 12 + p) & 63) * 1024);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

static double run(const char* src, int R, int D, int it_r, int it_d, unsigned* out, int mfma) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(64 * (R + D)), 144 * 1024, 0, src, R, D, it_r / 10 + 1, it_d / 10 + 1, out, mfma);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(64 * (R + D)), 144 * 1024, 0, src, R, D, it_r, it_d, out, mfma);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  char* src; unsigned* out;
  const size_t bytes = 4u << 20;
  (void)hipMalloc(&src, bytes); (void)hipMalloc(&out, 4096);
  std::vector<unsigned> h(bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i * 2654435761u;
  (void)hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  const int IR = 20000, ID = 2000;
  for (int mfma = 0; mfma < 2; ++mfma) {
    printf("readers %s\n", mfma ? "feed 8 MFMAs (32x32x16) per 8 reads: the GEMM's ratio is 6 reads per 8 MFMAs" : "only read");
    for (int R : {4, 8}) {
      const double tr = run(src, R, 0, IR, 0, out, mfma);
      printf("  %d readers alone: %.1f B/clk/CU (%.3f ms)\n", R, (double)R * IR * 8 * 1024 / tr / 2.4e9, tr * 1e3);
      for (int D : {4}) {
        const double td = run(src, 0, D, 0, ID, out, mfma);
        printf("  %d loaders alone: %.1f B/clk/CU (%.3f ms)\n", D, (double)D * ID * 12 * 1024 / td / 2.4e9, td * 1e3);
        // together: size the loaders' loop so that both sides would finish at about the same time if they did not interfere
        const int idm = (int)(ID * tr / td);
        const double tt = run(src, R, D, IR, idm, out, mfma);
        printf("  %d readers + %d loaders (%d loader iterations, sized to end together): %.3f ms  = %.2f x the readers alone; readers %.1f B/clk, loaders %.1f B/clk\n",
               R, D, idm, tt * 1e3, tt / tr, (double)R * IR * 8 * 1024 / tt / 2.4e9, (double)D * idm * 12 * 1024 / tt / 2.4e9);
      }
    }
  }
  return 0;
}
