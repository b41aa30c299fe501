// Micro-reproducer for the plane-sweep hazard of DESIGN.md section 5c ("skipping planes whose samples fall outside the partner
// image"): hand-counted `s_waitcnt vmcnt(N)` in front of loads that stay in flight across loop iterations.
//
// The producers of conv0_sweep.hip keep 16 gathers (4 chunks x 4 corners) in flight: chunk k of plane z + 1 is requested right
// after chunk k of plane z has been consumed, and every consumption is guarded by `s_waitcnt vmcnt(12)` — "all but the 12 youngest
// loads have returned", which names the 4 oldest (= this chunk) ONLY while exactly 16 loads are outstanding.  vmcnt counts what is
// outstanding, not which loads: when a wave does not issue the requests of a plane (a wave-uniform condition: the whole plane lies
// outside the partner image), 12 or fewer loads are outstanding at the following guards and `vmcnt(12)` is satisfied at once,
// although the chunk it is meant to guard has not landed — the consumer reads the registers' previous contents.
// RULE: a counted wait is valid for one issue history; every arm of a conditional issue needs its own counts (12 / 8 / 4 / 0 in
// the arm that issues nothing), or the skipped requests must still be issued (e.g. every lane from one address).
//
// The ring lives in ONE asm block with fixed registers: written in C++ with tied asm operands (as the sweep kernels are), hipcc
// re-homes the landing registers at the joins of such a branch and reuses them as temporaries under the landing loads — the first
// version of this file died of a memory fault from a clobbered offset, the OTHER mechanism of section 5b that
// tools/check_asm_gathers.py guards against; the two must be separated to see the counting rule alone.
// Variants: 0 = every plane requested, static counts; 1 = planes of a per-wave mask not requested, static counts (broken by
// construction); 2 = the same with per-arm counts.  Every load misses the caches (1 GiB buffer, 1 MiB stride).
// Build: hipcc --offload-arch=gfx950 -O3 -o vmcnt_conditional vmcnt_conditional.hip ; run: ./vmcnt_conditional [planes<=64] [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LSTRIDE 1052672        /* 1 MiB + 4 KiB between consecutive requests of a lane */
#define OFFMASK 0x3ffffff0     /* 1 GiB buffer of 16-byte records */
#define STR2(x) #x
#define STR(x) STR2(x)

__global__ void fill_kernel(uint4* p, unsigned nrec) {
  const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nrec) p[r] = make_uint4(r, ~r, r * 2654435761u, 0x5eed0000u | (r & 0xffffu));
}

// Register plan of the asm block (nothing in it is left to the register allocator):
//   v0 byte offset of this lane inside a request (lane * 16 + wave * 4096), v2 wrong-consumption count, v4..v7 temporaries,
//   v[16 + 16 k + 4 q .. +3] landing registers of chunk k, corner q;  s20 plane z, s21 planes, s[22:23] skip mask (bit z: no
//   requests for plane z), s24..s27 temporaries, s28 = the current plane's requests are in flight, s26 = plane z + 1 gets requested,
//   s29 = planes for which requests exist
#define REQ(K, Q)                                                                                                        \
  "s_add_u32 s24, s20, 1\n s_lshl_b32 s24, s24, 4\n s_add_u32 s24, s24, " STR(K) "*4+" STR(Q) "\n"                       \
  "s_mul_i32 s24, s24, " STR(LSTRIDE) "\n v_add_u32 v4, s24, v0\n v_and_b32 v4, " STR(OFFMASK) ", v4\n"                  \
  "global_load_dwordx4 v[16+16*" STR(K) "+4*" STR(Q) ":19+16*" STR(K) "+4*" STR(Q) "], v4, %[src]\n"
#define REQ4(K) REQ(K, 0) REQ(K, 1) REQ(K, 2) REQ(K, 3)
#define CHK(K, Q)                                                                                                        \
  "s_lshl_b32 s24, s20, 4\n s_add_u32 s24, s24, " STR(K) "*4+" STR(Q) "\n s_mul_i32 s24, s24, " STR(LSTRIDE) "\n"        \
  "v_add_u32 v4, s24, v0\n v_and_b32 v4, " STR(OFFMASK) ", v4\n v_lshrrev_b32 v4, 4, v4\n v_not_b32 v5, v4\n"            \
  "v_cmp_ne_u32 vcc, v4, v[16+16*" STR(K) "+4*" STR(Q) "]\n v_cndmask_b32 v6, 0, 1, vcc\n"                              \
  "v_cmp_ne_u32 vcc, v5, v[17+16*" STR(K) "+4*" STR(Q) "]\n v_cndmask_b32 v7, 0, 1, vcc\n"                              \
  "v_or_b32 v6, v6, v7\n v_add_u32 v2, v2, v6\n"
#define CHK4(K) CHK(K, 0) CHK(K, 1) CHK(K, 2) CHK(K, 3)
#define WAIT(N) "s_waitcnt vmcnt(" #N ")\n"
#define CLOBBER_V(a) "v" #a
#define RING_ASM(W1, W2, W3)                                                                                             \
  asm volatile(                                                                                                          \
      "v_mov_b32 v0, %[off0]\n v_mov_b32 v2, 0\n s_mov_b32 s21, %[planes]\n s_mov_b32 s29, %[preq]\n s_mov_b64 s[22:23], %[mask]\n"                \
      "s_mov_b32 s20, -1\n" REQ4(0) REQ4(1) REQ4(2) REQ4(3) "s_mov_b32 s20, 0\n s_mov_b32 s28, 1\n"                       \
      "1:\n"                                                                                                             \
      "s_add_u32 s25, s20, 1\n s_lshr_b64 s[26:27], s[22:23], s25\n s_and_b32 s26, s26, 1\n s_xor_b32 s26, s26, 1\n"      \
      "s_cmp_lt_u32 s25, s29\n s_cselect_b32 s26, s26, 0\n"                                                              \
      "s_cmp_eq_u32 s28, 0\n s_cbranch_scc1 3f\n"                                                                        \
      "s_cmp_eq_u32 s26, 0\n s_cbranch_scc1 2f\n"                                                                        \
      WAIT(12) CHK4(0) REQ4(0) WAIT(12) CHK4(1) REQ4(1) WAIT(12) CHK4(2) REQ4(2) WAIT(12) CHK4(3) REQ4(3)                 \
      "s_branch 4f\n"                                                                                                    \
      "2:\n" WAIT(12) CHK4(0) WAIT(W1) CHK4(1) WAIT(W2) CHK4(2) WAIT(W3) CHK4(3)                                          \
      "s_branch 4f\n"                                                                                                    \
      "3:\n s_cmp_eq_u32 s26, 0\n s_cbranch_scc1 4f\n" REQ4(0) REQ4(1) REQ4(2) REQ4(3)                                    \
      "4:\n s_mov_b32 s28, s26\n s_add_u32 s20, s20, 1\n s_cmp_lt_u32 s20, s21\n s_cbranch_scc1 1b\n"                     \
      "s_waitcnt vmcnt(0)\n v_mov_b32 %[wrong], v2\n"                                                                    \
      : [wrong] "=v"(wrong)                                                                                              \
      : [off0] "v"(off0), [planes] "s"(planes), [preq] "s"(preq), [mask] "s"(mask), [src] "s"(src)                                         \
      : "memory", "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "v0", "v2", "v4", "v5", "v6", "v7",   \
        "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33",   \
        "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",   \
        "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",   \
        "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79")

template <int VARIANT>
__global__ __launch_bounds__(64) void ring_kernel(const unsigned char* __restrict__ src, int planes, unsigned long long skipmask,
                                                  unsigned* __restrict__ bad) {
  const unsigned off0 = threadIdx.x * 16u + blockIdx.x * 4096u;
  // per-wave skip pattern (plane 0 is always requested); variant 0 skips nothing
  unsigned long long m = skipmask ^ ((unsigned long long)(blockIdx.x + 1) * 0x9E3779B97F4A7C15ull);
  const unsigned long long mask = VARIANT == 0 ? 0ull : (m & (m >> 1)) & ~1ull;        // about a quarter of the planes
  // variant 0 keeps the issue history static to the end: the last plane re-requests (a plane past the end, never consumed), as the
  // sweep kernels do ("a harmless re-request keeps the wait counts static")
  const int preq = VARIANT == 0 ? planes + 1 : planes;
  unsigned wrong = 0;
  if (VARIANT == 2) RING_ASM(8, 4, 0); else RING_ASM(12, 12, 12);
  if (wrong) atomicAdd(bad, wrong);
}

int main(int argc, char** argv) {
  const unsigned nrec = 1u << 26;
  int planes = argc > 1 ? atoi(argv[1]) : 48;
  if (planes > 64) planes = 64;
  const int waves = argc > 2 ? atoi(argv[2]) : 16384;
  unsigned char* d; unsigned* bad;
  if (hipMalloc(&d, (size_t)nrec * 16) != hipSuccess || hipMalloc(&bad, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipLaunchKernelGGL(fill_kernel, dim3(nrec / 256), dim3(256), 0, 0, reinterpret_cast<uint4*>(d), nrec);
  if (hipDeviceSynchronize() != hipSuccess) { printf("fill failed\n"); return 1; }
  const char* names[3] = {"every plane requested, static vmcnt(12)", "conditional issue, static vmcnt(12)  [broken by construction]",
                          "conditional issue, per-arm counts 12/8/4/0"};
  for (int v = 0; v < 3; ++v) {
    (void)hipMemset(bad, 0, 4);
    const unsigned long long sm = 0xA5A5F00F3C3C9669ull;
    if (v == 0) hipLaunchKernelGGL(ring_kernel<0>, dim3(waves), dim3(64), 0, 0, d, planes, sm, bad);
    if (v == 1) hipLaunchKernelGGL(ring_kernel<1>, dim3(waves), dim3(64), 0, 0, d, planes, sm, bad);
    if (v == 2) hipLaunchKernelGGL(ring_kernel<2>, dim3(waves), dim3(64), 0, 0, d, planes, sm, bad);
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("variant %d: %s\n", v, hipGetErrorString(e)); return 1; }
    unsigned hb = 0;
    (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("variant %d (%s): %u wrong 16-byte consumptions (of at most %llu)\n", v, names[v], hb, (unsigned long long)waves * 64ull * planes * 16ull);
  }
  (void)hipFree(d); (void)hipFree(bad);
  return 0;
}
