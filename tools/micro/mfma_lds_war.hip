// Does a ds_read that is issued right behind a burst of MFMAs land in the MFMAs' source registers before they have read them?
// (A candidate explanation of conv0_sweep_x3.hip's first consumer loop and of hazard (3) of the 16-bit sweep in DESIGN.md: about
// one consumer wave-plane in a thousand came out wrong until the loop was scheduled by hand.)  RESULT on MI355X / ROCm 7.2:
// no — B-operand reload, accumulator-input reload and reload of the input of a dependent chain's last MFMA are all exact for
// 1..16 MFMAs in flight, 0..128 wait states of gap, with and without the stress waves.  This candidate is ruled out.
//
// One measuring wave per workgroup: B = all-ones operand read from LDS; NM independent v_mfma_f32_16x16x32_bf16 (distinct
// accumulators, same A and B), then `s_nop` x GAP, then ds_read_b128 of an all-ZERO row into B's registers.  Every
// accumulator must end as 32.0 (K = 32 products of 1 x 1); one that read B after the reload landed ends as 0.
// Variants: STRESS = 1 adds 4 more waves to the workgroup (so one shares the measuring wave's SIMD) that hammer VALU +
// ds_write_b128, like the producer waves of the sweep kernels.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_lds_war.hip -o tools/micro/mfma_lds_war && ./tools/micro/mfma_lds_war
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int NM, int GAP, int STRESS, int SRCC>
__global__ __launch_bounds__(64 * (1 + 4 * STRESS)) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2 * 256 + 4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // row P: bf16 ones (SRCC: fp32 ones, the accumulator input), row Q: zeros
  for (int i = tid; i < 256; i += blockDim.x) { lds[i] = SRCC ? 0x3f800000u : 0x3f803f80u; lds[256 + i] = 0u; }
  __syncthreads();
  if (wave > 0) {
    // stress waves: VALU + LDS writes for the whole run
    float x = (float)tid;
    u4 v = {1u, 2u, 3u, 4u};
    for (int it = 0; it < iters * 40; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x = x * 1.0001f + 0.5f;
      v.x = __float_as_uint(x);
      *reinterpret_cast<u4*>(&lds[512 + ((tid * 4 + (it & 3) * 1024) & 4095)]) = v;
    }
    if (x == 12345.f) out[0] = x;
    return;
  }
  const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds + lane * 16;
  u4 A = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, B = {0u, 0u, 0u, 0u};
  int bad = 0, first_bad = -1;
  float sample = -1.f;
  for (int it = 0; it < iters; ++it) {
    f4 acc[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(B) : "v"(base) : "memory");        // B = ones
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    if (SRCC == 2) {  // dependent chain with rotating registers: acc[i] = A x A + acc[i-1]; the reload targets acc[NM-2], the
                      // accumulator INPUT of the last MFMA, which cannot be read before MFMA NM-2 has produced it
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %2" : "=v"(acc[0]) : "v"(A), "v"(B));
#pragma unroll
      for (int i = 1; i < NM; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %2" : "=&v"(acc[i]) : "v"(A), "v"(acc[i - 1]));
      if (GAP > 0) asm volatile(".rept %0\n\ts_nop 0\n\t.endr" :: "n"(GAP) : "memory");
      if (GAP >= 0) asm volatile("ds_read_b128 %0, %1 offset:1024" : "+v"(acc[NM - 2]) : "v"(base) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      asm volatile("" : "+v"(acc[NM - 1]));
      const float want = 32.0f * NM + 1.0f;
      if (acc[NM - 1].x != want || acc[NM - 1].w != want) { ++bad; if (first_bad < 0) first_bad = NM - 1; }
      sample = acc[NM - 1].x;
      continue;
    } else if (SRCC) {       // B (from LDS, fp32 ones) is the ACCUMULATOR INPUT of every MFMA; the product operands are A x A
#pragma unroll
      for (int i = 0; i < NM; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %2" : "=v"(acc[i]) : "v"(A), "v"(B));
    } else {
#pragma unroll
      for (int i = 0; i < NM; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
    }
    if (GAP > 0) asm volatile(".rept %0\n\ts_nop 0\n\t.endr" :: "n"(GAP) : "memory");
    if (GAP >= 0) asm volatile("ds_read_b128 %0, %1 offset:1024" : "+v"(B) : "v"(base) : "memory");        // B <- zeros (GAP < 0: control, no reload)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(B) :: "memory");
    // the accumulators come out of inline asm: the compiler knows neither that they are MFMA results (no hazard wait states) nor
    // that the wait above concerns them — every later read is made to depend on this statement
#pragma unroll
    for (int i = 0; i < NM; ++i) asm volatile("" : "+v"(acc[i]));
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const float want = SRCC ? 33.0f : 32.0f;
      if (acc[i].x != want || acc[i].w != want) { ++bad; if (first_bad < 0) first_bad = i; }
      if (i == NM - 1) sample = acc[i].x;
    }
  }
  out[(blockIdx.x * 64 + lane) * 2] = (float)bad;
  out[(blockIdx.x * 64 + lane) * 2 + 1] = (float)first_bad;
  if (blockIdx.x == 0 && lane == 0) out[1024 * 64 * 2] = sample;
}

template <int NM, int GAP, int STRESS, int SRCC = 0>
void run(float* d, int blocks, int iters) {
  hipMemset(d, 0, blocks * 64 * 2 * sizeof(float));
  hipLaunchKernelGGL((k<NM, GAP, STRESS, SRCC>), dim3(blocks), dim3(64 * (1 + 4 * STRESS)), 0, 0, d, iters);
  std::vector<float> h(blocks * 64 * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  double bad = 0; int lanes = 0, fb = 99;
  for (int i = 0; i < blocks * 64; ++i) { bad += h[2 * i]; if (h[2 * i] > 0) { ++lanes; if ((int)h[2 * i + 1] < fb) fb = (int)h[2 * i + 1]; } }
  float smp = 0; hipMemcpy(&smp, d + 1024 * 64 * 2, 4, hipMemcpyDeviceToHost);
  printf("%-12s MFMAs %2d  gap %3d wait states  stress %d:  corrupted accumulators %.0f of %.0f (lanes affected %d, earliest MFMA index %d)  [sample acc %g]\n", SRCC == 2 ? "chain SrcC" : SRCC ? "SrcC reload" : "SrcB reload", NM, GAP, STRESS,
         bad, (double)blocks * 64 * iters * (SRCC == 2 ? 1 : NM), lanes, lanes ? fb : -1, smp);
}

int main() {
  float* d; hipMalloc(&d, (1024 * 64 * 2 + 4) * sizeof(float));
  const int blocks = 1024, iters = 200;
  run<12, -1, 0>(d, blocks, iters); run<12, -1, 1>(d, blocks, iters);
  run<1, 0, 0>(d, blocks, iters);  run<4, 0, 0>(d, blocks, iters);  run<8, 0, 0>(d, blocks, iters);  run<12, 0, 0>(d, blocks, iters);  run<16, 0, 0>(d, blocks, iters);
  run<12, 16, 0>(d, blocks, iters); run<12, 32, 0>(d, blocks, iters); run<12, 64, 0>(d, blocks, iters); run<12, 128, 0>(d, blocks, iters);
  run<1, 0, 1>(d, blocks, iters);  run<4, 0, 1>(d, blocks, iters);  run<8, 0, 1>(d, blocks, iters);  run<12, 0, 1>(d, blocks, iters);  run<16, 0, 1>(d, blocks, iters);
  run<12, 16, 1>(d, blocks, iters); run<12, 32, 1>(d, blocks, iters); run<12, 64, 1>(d, blocks, iters); run<12, 128, 1>(d, blocks, iters);
  run<12, -1, 0, 1>(d, blocks, iters);
  run<1, 0, 0, 1>(d, blocks, iters); run<2, 0, 0, 1>(d, blocks, iters); run<4, 0, 0, 1>(d, blocks, iters); run<8, 0, 0, 1>(d, blocks, iters); run<12, 0, 0, 1>(d, blocks, iters); run<16, 0, 0, 1>(d, blocks, iters);
  run<12, 16, 0, 1>(d, blocks, iters); run<12, 32, 0, 1>(d, blocks, iters); run<12, 64, 0, 1>(d, blocks, iters); run<12, 128, 0, 1>(d, blocks, iters);
  run<4, 0, 1, 1>(d, blocks, iters); run<12, 0, 1, 1>(d, blocks, iters); run<12, 32, 1, 1>(d, blocks, iters); run<12, 64, 1, 1>(d, blocks, iters); run<12, 128, 1, 1>(d, blocks, iters);
  run<8, -1, 0, 2>(d, blocks, iters);
  run<2, 0, 0, 2>(d, blocks, iters); run<4, 0, 0, 2>(d, blocks, iters); run<8, 0, 0, 2>(d, blocks, iters); run<12, 0, 0, 2>(d, blocks, iters);
  run<8, 4, 0, 2>(d, blocks, iters); run<8, 16, 0, 2>(d, blocks, iters); run<8, 64, 0, 2>(d, blocks, iters); run<8, 0, 1, 2>(d, blocks, iters); run<12, 0, 1, 2>(d, blocks, iters);
  return 0;
}
