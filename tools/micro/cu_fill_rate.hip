// Micro-benchmark: how many bytes per clock can ONE CU pull out of L2 — as LDS-DMA (global_load_lds_dwordx4), as plain 16-byte loads
// into VGPRs, as VGPR loads + ds_write_b128 (register staging), or as a mix of LDS-DMA and register staging — in the access shape of
// the implicit GEMM's request waves: a "piece" is one wave instruction = 8 rows x 128 bytes (lane (r8, j) reads chunk j of row r8),
// rows `stride` bytes apart, all workgroups of an XCD walking the same L2-resident region.  One workgroup per CU, `waves` loader
// waves, 12 pieces in flight per wave (counted vmcnt).  Round 5: the K loop of conv_igemm_ws_kernel turned out to be bound by the
// request waves' issue back-pressure (1800 cycles per 48 KB K tile), i.e. by this number, not by the matrix pipe.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/cu_fill_rate tools/micro/cu_fill_rate.hip ; run: tools/micro/cu_fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_off) : "memory");
}

__device__ __forceinline__ void blds16(unsigned voff, const __amdgpu_buffer_rsrc_t& rsrc, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds_off) : "memory");
}
// MODE 4: LDS-DMA through a buffer descriptor (SGPR base + 32-bit per-lane offset); MODE 5: as 0 with a second batch of 12 in flight (vmcnt(12))
// MODE 0: LDS-DMA; 1: VGPR loads (discarded); 2: VGPR loads + ds_write_b128; 3: pieces alternate DMA (2 of 3) and register staging (1 of 3)
template <int MODE>
__global__ __launch_bounds__(512) void fill(const char* src, size_t region, int stride, int iters, float* out) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int j = lane & 7, r8 = lane >> 3;
  const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds + wave * 12 * 1024;
  // all workgroups of an XCD (blockIdx & 7) walk the same region; a piece covers 8 rows
  const unsigned rmask = (unsigned)(region / (size_t)stride) - 1u;      // rows: a power of two
  unsigned row = ((unsigned)(blockIdx.x >> 3) * 40u + wave * 8u) & rmask;
  u32x4 v[12];
  unsigned acc = 0;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)0x7fffffff, 0x00020000);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 12; ++p) {
      const char* a = src + (size_t)((row + r8) & rmask) * (size_t)stride + j * 16;
      const unsigned ao = ((row + r8) & rmask) * (unsigned)stride + j * 16;
      row = (row + 8 * nw) & rmask;
      const bool dma = MODE == 0 || MODE == 5 || (MODE == 3 && (p % 3) != 2);
      if (MODE == 6) {      // 16 rows x 64 bytes per piece: lane (r16, j4) reads chunk j4 of row r16 (two such pieces cover the 128-byte lines of 16 rows)
        const unsigned ao6 = ((row + (p & 1 ? 8 : 0) * 0 + (lane >> 2) + (p >> 1 << 4)) & rmask) * (unsigned)stride + (p & 1) * 64 + (lane & 3) * 16;
        blds16(ao6, rsrc, lbase + p * 1024);
      } else if (MODE == 4) blds16(ao, rsrc, lbase + p * 1024);
      else if (dma) glds16(a, lbase + ((MODE == 5 && (it & 1)) ? 12 * 1024 * 8 : 0) + p * 1024);
      else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[p]) : "v"(a) : "memory");
    }
    // wait for everything (the request waves of the GEMM keep one more step in flight; this is the simplest steady state with 12 in flight)
    if (MODE == 5) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (MODE == 0 || MODE == 4 || MODE == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (MODE == 1) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]) :: "memory");
#pragma unroll
      for (int p = 0; p < 12; ++p) acc ^= v[p][0];
    } else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]) :: "memory");
#pragma unroll
      for (int p = 0; p < 12; ++p) {
        const bool dma = MODE == 3 && (p % 3) != 2;
        if (!dma) asm volatile("ds_write_b128 %0, %1" :: "v"(lbase + p * 1024 + lane * 16), "v"(v[p]) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (acc == 0x12345678u || lds[threadIdx.x].x == 0x9abcdef0u) out[blockIdx.x] = 1.f;
}

template <int MODE>
static double run(const char* src, size_t region, int stride, int waves, int iters, float* out) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(fill<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(fill<MODE>, dim3(256), dim3(64 * waves), 150 * 1024, 0, src, region, stride, iters / 10 + 1, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fill<MODE>, dim3(256), dim3(64 * waves), 150 * 1024, 0, src, region, stride, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)iters * 12 * 1024 * waves / (ms * 1e-3);      // bytes per second per CU
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  char* src; float* out;
  const size_t bytes = 256u << 20;
  hipMalloc(&src, bytes + 4096); hipMalloc(&out, 4096);
  std::vector<unsigned> h(bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i * 2654435761u;
  hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
  const char* names[6] = {"LDS-DMA", "VGPR loads", "VGPR + ds_write", "2/3 DMA + 1/3 staged", "buffer LDS-DMA", "LDS-DMA 24 in flight"};
  printf("bytes per second per CU (GB/s) [B/clk at 2.4 GHz]; 256 workgroups, 12 pieces of 1 KiB in flight per wave\n");
  for (size_t region : {(size_t)2 << 20}) {      // 1 MB: L2-resident; 16 MB: L2 + infinity cache
    for (int stride : {1024, 8192}) {
      for (int waves : {1, 2, 3, 4, 6, 8}) {
        printf("region %2zu MB stride %5d waves %d:", region >> 20, stride, waves);
        double r;
        r = run<0>(src, region, stride, waves, iters, out); printf("  %s %6.1f [%4.1f]", names[0], r / 1e9, r / 2.4e9);
        r = run<1>(src, region, stride, waves, iters, out); printf("  %s %6.1f [%4.1f]", names[1], r / 1e9, r / 2.4e9);
        r = run<4>(src, region, stride, waves, iters, out); printf("  %s %6.1f [%4.1f]", names[4], r / 1e9, r / 2.4e9);
        r = run<6>(src, region, stride, waves, iters, out); printf("  buffer 16x64B %6.1f [%4.1f]", r / 1e9, r / 2.4e9);
        if (waves <= 4) { r = run<5>(src, region, stride, waves, iters, out); printf("  %s %6.1f [%4.1f]", names[5], r / 1e9, r / 2.4e9); }
        printf("\n");
        fflush(stdout);
      }
    }
  }
  return 0;
}
