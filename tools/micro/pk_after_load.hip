// Stand-alone attempt at the round-4 hazard: a packed fp32 instruction that consumes registers a global_load has just delivered,
// behind the correct s_waitcnt, while another stream's kernel shares the CUs.  Kernel A, per lane: pre-fills v[4:7] with a marker, loads a
// 48-byte record with three global_load_dwordx4 (every lane of a wave the same record, a different one per wave: cache-cold) plus one more dword,
// waits with the counted vmcnt hipcc used in fuse_points_kernel, and forms   pk = v_pk_mul_f32(rec[0:1], w)   immediately; after all loads have
// landed and 32 wait states it forms the same two products with plain v_mul_f32 from the same registers.  Any lane whose packed products differ
// from the plain ones is counted per lane row.  Co-tenant on another stream: 0 none, 1 LDS + MFMA kernel, 2 MFMA stream.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_after_load pk_after_load.hip ; run: ./pk_after_load
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void kernel_a(const float* __restrict__ table, const float* __restrict__ extra, unsigned* __restrict__ bad, int nrec) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const float* rec = table + (size_t)((wave * 2654435761u) % (unsigned)nrec) * 12;      // one record per wave, scattered over 1.5 GB
  const float* ex = extra + (threadIdx.x & 63);
  float wx = 1.0f + (float)(threadIdx.x & 7), wy = 2.0f + (float)(threadIdx.x & 3);
  float pk0, pk1, pl0, pl1;
  asm volatile(
      "v_mov_b32 v4, 0x7fc00001\n v_mov_b32 v5, 0x7fc00002\n v_mov_b32 v6, 0x7fc00003\n v_mov_b32 v7, 0x7fc00004\n"
      "v_mov_b32 v22, %[wx]\n v_mov_b32 v23, %[wy]\n"
      "global_load_dwordx4 v[4:7], %[rec], off\n"
      "global_load_dwordx4 v[28:31], %[rec], off offset:16\n"
      "global_load_dwordx4 v[42:45], %[rec], off offset:32\n"
      "global_load_dword v8, %[ex], off\n"
      "s_waitcnt vmcnt(3)\n"
      "v_pk_mul_f32 v[10:11], v[4:5], v[22:23]\n"
      "s_waitcnt vmcnt(0)\n"
      "s_nop 15\n s_nop 15\n"
      "v_mul_f32 v12, v4, v22\n v_mul_f32 v13, v5, v23\n"
      "v_mov_b32 %[pk0], v10\n v_mov_b32 %[pk1], v11\n v_mov_b32 %[pl0], v12\n v_mov_b32 %[pl1], v13\n"
      : [pk0] "=v"(pk0), [pk1] "=v"(pk1), [pl0] "=v"(pl0), [pl1] "=v"(pl1)
      : [rec] "v"(rec), [ex] "v"(ex), [wx] "v"(wx), [wy] "v"(wy)
      : "memory", "v4", "v5", "v6", "v7", "v8", "v10", "v11", "v12", "v13", "v22", "v23", "v28", "v29", "v30", "v31", "v42", "v43", "v44", "v45");
  if (__float_as_uint(pk0) != __float_as_uint(pl0) || __float_as_uint(pk1) != __float_as_uint(pl1)) atomicAdd(bad + (threadIdx.x & 63), 1u);
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

__global__ __launch_bounds__(256) void kernel_stream(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {      // memory-bound co-tenant
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  const int nrec = 32 * 1024 * 1024;
  float *table, *extra, *sink; unsigned* bad; uint4 *cs, *cd;
  if (hipMalloc(&table, (size_t)nrec * 48) != hipSuccess || hipMalloc(&extra, 256) != hipSuccess || hipMalloc(&bad, 256) != hipSuccess ||
      hipMalloc(&sink, 8192 * 256 * 4) != hipSuccess || hipMalloc(&cs, 1u << 30) != hipSuccess || hipMalloc(&cd, 1u << 30) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(table, 0x3f, (size_t)nrec * 48); hipMemset(extra, 0, 256);
  hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
  const char* names[3] = {"kernel A alone", "kernel A next to an LDS + MFMA kernel (other stream)", "kernel A next to a copy kernel (other stream)"};
  for (int mode = 0; mode < 3; ++mode) {
    unsigned total = 0, rows[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 20; ++rep) {
      hipMemset(bad, 0, 256);
      hipDeviceSynchronize();
      if (mode == 1) hipLaunchKernelGGL(kernel_lds_mfma, dim3(4096), dim3(256), 25600, sb, sink, 20000);
      if (mode == 2) hipLaunchKernelGGL(kernel_stream, dim3(4096), dim3(256), 0, sb, cs, cd, (size_t)(1u << 30) / 16);
      for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(kernel_a, dim3(8192), dim3(256), 0, sa, table, extra, bad, nrec);
      hipDeviceSynchronize();
      unsigned hb[64];
      hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l) { total += hb[l]; rows[l / 16] += hb[l]; }
    }
    printf("%-52s: %u lanes with packed != plain products (lane rows 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u) of %llu\n", names[mode], total, rows[0], rows[1],
           rows[2], rows[3], 20ull * 8 * 8192 * 256);
  }
  return 0;
}
