// conv0 of the cost-regularisation net (Conv3d 32->8, k3 p1 + BN + ReLU, network_v5.py:260-291) with the plane sweep
// (homo_warping + "ref + warped" fusion, network_v5.py:378-430) built on the fly — depth-sweeping, role-specialised
// 16-bit kernels for gfx950.  They replace the halo-tile kernel (conv3d_tile.hip, layer 10) whose staging, MFMA and store
// phases ran back to back: all blocks of a launch march through identical phases in lock step, so nothing overlaps.
//
// A 12x16 (H x W) column of one view is swept through all D depth planes by one workgroup:
//   * producer waves (4, one per SIMD): per plane they project the 14x18 input plane's voxels (incl. halo) with that plane's depth
//     (sw_ray / sw_corner_weights), gather the 4 bilinear corners of the partner view's feature map (corners outside the image get
//     weight 0: grid_sample padding_mode="zeros"), blend, and write the 64-byte voxels into an LDS ring of plane slots.  The corner
//     loads of plane z+1 are issued (inline asm, counted vmcnt) right behind the blend of plane z out of the same registers, so
//     gather latency is covered.  Halo redundancy is 252/192 = 1.31x (the 4x8x8 tile: 2.34x).
//   * consumer waves (4 x 3 tile rows; 3 x 4 in rounds 1-4): MFMA 16x16x32, input-plane stationary.  Cout = 8 fills only half of the 16
//     MFMA rows, so two depth taps share one instruction: A01[t] = rows 0-7 W(kd=0,t), rows 8-15 W(kd=1,t); A2[t] = rows 8-15
//     W(kd=2,t).  For input plane p and in-plane tap t:  X[p] += A01[t]*B,  X[p-1] += A2[t]*B  (same B register), hence
//     out[o] = rows 8-15 of X[o] (kd=1 from plane o, kd=2 from plane o+1)  +  rows 0-7 of X[o-1] (kd=0 from plane o-1).
//     18 MFMAs per 16-voxel fragment and plane instead of 27, weights live in registers, the folded BN shift is the accumulators'
//     initial value.  The two row halves sit in lanes 0-31 / 32-63 of the accumulator: v_permlane32_swap pairs two fragments so the
//     final add, ReLU and the 8-byte stores run on all 64 lanes.
// One s_barrier per plane hands a ring slot from the producers to the consumers; producers fill the next slot meanwhile.
//
// Three forms (launch_conv0_sweep picks):
//   conv0_sweep_kernel<bf16, bf16, 0/1/2>   bf16 feature map, one lane per voxel, fp32 blend (rounds 1-4; option sweep_f16 = 0)
//   conv0_sweep_kernel<f16, f16, 0/3>       fp16 nets: fp32-accumulating v_fma_mix blend (0) or packed f16 (3, debug flag 2097152)
//   conv0_sweep_persistent_kernel<bf16>     bf16 nets since round 5: f16 feature map, packed-f16 blend, cooperative gathers (four lanes
//                                           per corner pixel), one workgroup per CU walking its tiles with both roles streaming across
//                                           tile boundaries; conv0_sweep_kernel<f16, bf16, 3> is its one-tile twin (debug flag 268435456),
//                                           bit-identical.  Why packed f16: tools/micro/mfma_valu_coissue.hip, DESIGN 5e.
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "prof.h"

// (Rounds 1-5 carried timing-build switches here - `SW_ABL`: no blend / no MFMA / no gathers / phase timers ... - whose measurements are in
// DESIGN.md sections 5, 5d and 5e; they left the source in round 6 together with the two-plane-lag variants `SW_PRE` / `SWP_PRE`.)

namespace rgbm {

namespace {

// Tile: SW_TH x 16 output voxels per plane.  TH = 12 makes the input plane incl. halo 14 x 18 = 252 voxels = four
// producer waves with 98 % of their lanes busy, ONE per SIMD: the producers are VALU-issue bound (a wave64 VALU
// instruction holds a SIMD's issue port for 4 cycles; measured, the first 16x16 version put two producer waves on two
// of the SIMDs and ran at their pace), so lanes per instruction and waves per SIMD are what set the speed.
constexpr int SW_TH = 12, SW_TW = 16;
constexpr int SW_HH = SW_TH + 2, SW_HW = SW_TW + 2;
constexpr int SW_NV = SW_HH * SW_HW;             // 252 voxels per input plane
#ifndef SW_COOP
#define SW_COOP 1     // packed-f16 instantiations: cooperative gathers - four lanes read the four 16-byte chunks of ONE corner pixel (a gather instruction
                      // touches 16 pixels' 64-byte runs instead of 64 lanes' 16-byte pieces of 64 pixels).  Same box, dense, ms per step:
                      // lane per voxel 14.05, quads with a DPP exchange 13.2 (and 15.4 with 16 consecutive voxels per round + ds_bpermute_b32)
#endif
// (Round 5 measured the consumers TWO planes behind the producers, reading the first operand fragments of the next plane during the current plane's
// output epilogue: it only moves the time - the MFMA phase shrinks by 100 cycles, the epilogue with the reads in it grows by as much: 13.2 ms dense
// without, 13.3 with six fragments.)
constexpr int SW_LAG = 1;                        // the consumers run one plane behind the producers
#ifndef SW_NSLOT_N
#define SW_NSLOT_N 3
#endif
#ifndef SW_VS_BYTES
#define SW_VS_BYTES 80
#endif
constexpr int SW_VS = SW_VS_BYTES;               // LDS bytes per voxel: 64 data + 16 pad (conflict-free b128 rows)
constexpr int SW_SLOT = SW_NV * SW_VS;           // 20160
constexpr int SW_NPW = (SW_NV + 63) / 64;        // 4 producer waves
#ifndef SW_SKIP
#define SW_SKIP 0     // 1 = bf16 producers: planes on which a wave's 64 voxels all project outside the partner image store the reference features
                      // unblended (round 4: stable in every gate once the library held no packed fp32, -2.4 % bf16 / +4 % f16 with the dot2 blend).
                      // Round 5, with four consumer waves and the fp32 blend, same box, two interleaved rounds (tools/kernel_ms.py): dense 16.7 ->
                      // 16.4 ms, but 10.11 -> 10.13 ms on the tiles the sparse cost regularisation needs (the planes that can be skipped lie in
                      // tiles that are skipped anyway); all 53 sweep / golden / stability / overlap tests green.  Off: nothing on the shipped path.
#endif
#ifndef SW_OCC
#define SW_OCC 3     // minimum waves per SIMD the register allocation must allow (3: 168 VGPRs, one 8-wave workgroup per CU plus the next one's early waves)
#endif
#ifndef SW_CR
#define SW_CR 3       // tile rows (= 16-voxel fragments) per consumer wave: 4 -> 3 consumer waves (rounds 1-4), 3 -> 4 (round 5, below), 2 -> 6.
                      // Measured alone (no producer work at all) the 4-row consumer needs 2140 cycles per plane for 1152 cycles of MFMA: one
                      // wave per SIMD has nobody to cover its LDS round trip at the start of a plane and its ~70 VALU of output epilogue at
                      // the end.  With 2 rows per wave two of the SIMDs host two consumer waves each, whose streams the hardware interleaves —
                      // measured in the full kernel: 17.0 ms against 16.9 ms (no gain: the consumers' bubbles are not what bounds it), and
                      // the f16 instantiation lost its run-to-run stability (test_fp16_sweep_conv0_stable_and_matches_tile_conv0).  3 rows
                      // (4 consumer waves, one per SIMD next to a producer: the balanced split): 16.2 ms against 16.8 ms in bf16, and the f16
                      // instantiation fails its golden test — with hipcc's packed fp32 instructions in the library.  Round 5, on the library
                      // without them (since the end of round 4): 3 rows pass every gate of both types (sweep / golden / stability, 30 tests; the
                      // 20-run at-batch determinism gate) and measure 16.7 -> 16.2 ms dense in bf16, 15.25 -> 15.07 in f16 (same box, two
                      // interleaved rounds, tools/kernel_ms.py): one consumer wave per SIMD next to its producer.  3 is the default now.
#endif
constexpr int SW_CR_ = SW_CR;
constexpr int SW_NCW = SW_TH / SW_CR_;           // consumer waves, SW_CR fragments (rows) each
constexpr int SW_THREADS = (SW_NPW + SW_NCW) * 64;
// Plane ring of THREE slots.  Two are what the barrier protocol needs on paper (producers fill slot z&1 while the consumers
// read the other one), and the bf16 build never showed a problem with two — but the f16_t build with the shorter
// v_fma_mix blend did: about one workgroup in a thousand delivered one consumer wave's rows of three consecutive output
// planes differently from run to run (= one input plane read while it was being overwritten), under every variation tried
// of wait counts, s_nops behind the LDS stores, volatile blend asm and FP16_OVFL on/off; any build whose producers were
// slower (explicit saturation code in the pack) was stable, and so is the third slot, which gives every plane one more
// barrier interval before its slot is reused (tools/f16_sweep_check.py 256).  60 KB per workgroup; occupancy is set by
// the VGPRs, not by LDS.
constexpr int SW_NSLOT = SW_NSLOT_N;
constexpr int SW_LDS = SW_NSLOT * SW_SLOT;
static_assert(SW_TH % SW_CR_ == 0, "a consumer wave owns SW_CR rows");
static_assert(SW_NSLOT_N > SW_LAG, "the ring holds the plane being written and the SW_LAG behind it");
constexpr int SW_NPAIR = (SW_CR_ + 1) / 2;       // fragment pairs of the output epilogue; with an odd SW_CR the last fragment pairs with itself

struct SweepDesc {
  const unsigned short* feat;     // [V][H][W][32] bf16
  const unsigned short* wgt;      // [18][16][4][8] bf16 (conv0_sweep_pack)
  const float* bias;              // [16] folded BN shift
  const float* homog;             // [V][12]
  const float* depths;            // [B][D]
  unsigned short* out;            // [N][D][H][W][8] bf16
  int N, D, H, W, v0, V, B, nth, ntw, relu;
  const int* tile_list; const int* tile_count;      // sparse cost regularisation: only these tiles (ascending), else null
};

struct Corner {                     // everything the blend of one plane needs besides the gathered data
  unsigned off[4];                  // byte offsets of the 4 (clamped) corner pixels inside the partner feature map
  float w[4];                       // bilinear weights: 0 for a corner outside the image, NaN for a non-finite projection
  unsigned wp[4];                   // blend mode 2: [0..1] = {bf16(w0) | bf16(w1) << 16, bf16(w2) | bf16(w3) << 16}; f16_t: [q] = f16(w[q]) in both halves
  bool skip;                        // SW_SKIP: every lane of the wave has four zero weights on this plane (wave-uniform)
};

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;     // native vector: usable as a tied inline-asm operand

template <typename T> struct Sw16;                                   // the two 16-bit storage types of this kernel
typedef __attribute__((ext_vector_type(4))) unsigned sw_u4v;       // native vector: usable as a tied inline-asm operand
template <> struct Sw16<unsigned short> {                            // bf16: a dword's halves are the high halves of two floats
  __device__ static __forceinline__ f32x2 unpack(unsigned u) { return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
  __device__ static __forceinline__ unsigned pack(f32x2 v) { return pack2_bf16(v.x, v.y); }
  __device__ static __forceinline__ f32x4 mma(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Sw16<f16_t> {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  __device__ static __forceinline__ f32x2 unpack(unsigned u) { return __builtin_convertvector(__builtin_bit_cast(h2, u), f32x2); }
  __device__ static __forceinline__ unsigned pack(f32x2 v) {
    const h2 h = {(f16_t)sat_f16(v.x), (f16_t)sat_f16(v.y)};
    return __builtin_bit_cast(unsigned, h);
  }
  __device__ static __forceinline__ f32x4 mma(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// Blend modes of the bf16 producers (template parameter BL of the kernel; rgbm_debug_flags 2097152 / 4194304 select 1 / 0 for A/B):
//   0  [default since round 5] fp32 arithmetic on unpacked channel pairs (written for v_pk_mul / v_pk_fma / v_pk_add_f32; the library is
//      built without packed fp32 instructions since round 4, so these are plain fp32 FMAs): 8 unpack + 5 packed + 1 convert per
//      dword.  Packed fp32 instructions next to another wave's MFMA stream on the same SIMD cost far more than their issue slot
//      (MI355X_MICROARCH.md: one v_pk_fma_f32 instead of two v_fma_f32 = +22 cycles) — and three of the four producers share
//      their SIMD with a consumer.
//   1  the same sum on scalar v_fma_f32 (inline asm: the SLP vectoriser would re-pack plain C): 8 unpack + 10 + 1 per dword,
//      bit-identical to mode 0.
//   2  [round-4 default; debug flag 4194304 since round 5] channel pairs stay packed: v_perm_b32 puts the low (high) halves of two corners' dwords side by side and
//      v_dot2_f32_bf16 multiplies them by the corner weights as a bf16 pair, accumulating in fp32 on top of the reference feature:
//      4 perm + 4 dot2 + 1 convert per dword, no packed-fp32 instruction.  The products are exact (bf16 x bf16 in fp32), the weights
//      carry 8 significant bits — the rounding the gathered features already have; a corner outside the image keeps weight 0 exactly,
//      a non-finite projection keeps NaN.
template <typename T, int BL>
__device__ __forceinline__ uint4 blend_chunk(const uint4& r, const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& e,
                                             const Corner& cn) {
  const float* w = cn.w;
  const unsigned rr[4] = {r.x, r.y, r.z, r.w}, aa[4] = {a[0], a[1], a[2], a[3]}, bb[4] = {b[0], b[1], b[2], b[3]};
  const unsigned cc[4] = {c[0], c[1], c[2], c[3]}, ee[4] = {e[0], e[1], e[2], e[3]};
  unsigned o[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {       // one dword = two 16-bit channels
    if constexpr (BL == 2) {
      const f32x2 fr = Sw16<T>::unpack(rr[q]);
      const unsigned ab_lo = __builtin_amdgcn_perm(bb[q], aa[q], 0x05040100u), ab_hi = __builtin_amdgcn_perm(bb[q], aa[q], 0x07060302u);
      const unsigned ce_lo = __builtin_amdgcn_perm(ee[q], cc[q], 0x05040100u), ce_hi = __builtin_amdgcn_perm(ee[q], cc[q], 0x07060302u);
      // (the builtin, not inline asm: a dot instruction's result needs three wait states before a VALU instruction of another
      // kind may read it on gfx940+, which hipcc only honours for instructions it can see — an asm version of these four lines
      // gave run-to-run different voxels)
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
      const bf2 w01 = __builtin_bit_cast(bf2, cn.wp[0]), w23 = __builtin_bit_cast(bf2, cn.wp[1]);
      float t0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ab_lo), w01, fr.x, false);
      float t1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ab_hi), w01, fr.y, false);
      t0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ce_lo), w23, t0, false);
      t1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, ce_hi), w23, t1, false);
      o[q] = Sw16<T>::pack(f32x2{t0, t1});
    } else if constexpr (BL == 1) {
      const f32x2 fr = Sw16<T>::unpack(rr[q]), fa = Sw16<T>::unpack(aa[q]), fb = Sw16<T>::unpack(bb[q]);
      const f32x2 fc = Sw16<T>::unpack(cc[q]), fe = Sw16<T>::unpack(ee[q]);
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float t;
        asm("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(fa[h]), "v"(w[0]));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(fb[h]), "v"(w[1]), "v"(t));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(fc[h]), "v"(w[2]), "v"(t));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(fe[h]), "v"(w[3]), "v"(t));
        asm("v_add_f32 %0, %1, %2" : "=v"(v[h]) : "v"(fr[h]), "v"(t));
      }
      o[q] = Sw16<T>::pack(f32x2{v[0], v[1]});
    } else {
      const f32x2 fr = Sw16<T>::unpack(rr[q]), fa = Sw16<T>::unpack(aa[q]), fb = Sw16<T>::unpack(bb[q]);
      const f32x2 fc = Sw16<T>::unpack(cc[q]), fe = Sw16<T>::unpack(ee[q]);
      const f32x2 v = fr + (((fa * w[0] + fb * w[1]) + fc * w[2]) + fe * w[3]);
      o[q] = Sw16<T>::pack(v);
    }
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// f16_t: the same sum with v_fma_mix_f32, which reads the f16 half of a dword directly (no v_cvt per operand): 5
// instructions per channel instead of 10 conversions + 5 packed operations per channel pair.  The caller has set
// MODE.FP16_OVFL, so the final v_cvt_pk_f16_f32 saturates at +-65504 by itself (NaN stays NaN) — sat_f16() in front of it
// costs 6 more instructions per dword.
#define SW_MIX(HI, D, H, W, C) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[" #HI ",0,0] op_sel_hi:[1,0,0]" : "=v"(D) : "v"(H), "v"(W), "v"(C))
template <bool PK>
__device__ __forceinline__ uint4 blend_chunk_f16(const uint4& r, const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& e,
                                                 const Corner& cn) {
  typedef Sw16<f16_t>::h2 h2;
  const float* w = cn.w;
  // (element copies: indexing the u32x4 references with the unrolled loop counter made hipcc use element 0 for all four)
  const unsigned rr[4] = {r.x, r.y, r.z, r.w}, aa[4] = {a[0], a[1], a[2], a[3]}, bb[4] = {b[0], b[1], b[2], b[3]};
  const unsigned cc[4] = {c[0], c[1], c[2], c[3]}, ee[4] = {e[0], e[1], e[2], e[3]};
  unsigned o[4];
  if constexpr (PK) {
  // Round 5: packed f16 FMAs with the weights as f16 pairs (cn.wp) - four instructions per dword and no conversion at the end.
  // tools/micro/mfma_valu_coissue.hip: a SIMD issues NO fp32 arithmetic (v_fma_f32, v_fma_mix_f32, v_dot2_f32_bf16, v_cvt_pk_*) of one
  // wave while another wave's MFMAs occupy the matrix pipe - the two times add - whereas v_pk_fma_f16, v_perm_b32 and integer
  // instructions fit between the MFMAs (about two per MFMA).
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned t;
    asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(t) : "v"(aa[q]), "v"(cn.wp[0]), "v"(rr[q]));
    asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(t) : "v"(bb[q]), "v"(cn.wp[1]));
    asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(t) : "v"(cc[q]), "v"(cn.wp[2]));
    asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(t) : "v"(ee[q]), "v"(cn.wp[3]));
    o[q] = t;
  }
  } else {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float lo, hi;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(aa[q]), "v"(w[0]));
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(aa[q]), "v"(w[0]));
    SW_MIX(0, lo, bb[q], w[1], lo); SW_MIX(1, hi, bb[q], w[1], hi);
    SW_MIX(0, lo, cc[q], w[2], lo); SW_MIX(1, hi, cc[q], w[2], hi);
    SW_MIX(0, lo, ee[q], w[3], lo); SW_MIX(1, hi, ee[q], w[3], hi);
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(rr[q]), "v"(lo));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(rr[q]), "v"(hi));
    const h2 h = {(f16_t)lo, (f16_t)hi};
    o[q] = __builtin_bit_cast(unsigned, h);
  }
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}
#undef SW_MIX

// Projection of a voxel's pixel onto the partner view and its bilinear corners - ONE definition with explicit FMAs for every form of the
// producers (lane per voxel, cooperative, persistent), so that they agree bit for bit (left to -ffp-contract, hipcc fused `a * x + b * y + c`
// differently in two of them: c0 of the persistent kernel differed from the one-tile kernel's in 1 % of its elements by one bf16 step).
// ray = rot * (x, y, 1); per plane p = ray * depth + t, one reciprocal, and the reference's align_corners=True normalisation + grid_sample's
// align_corners=False un-normalisation folded into ix = u * W / (W - 1) - 0.5 (sx, sy).  Algebraically network_v5.py:378-430; rounding
// differs from the fp32 kernels in the last ulps of the coordinate (1e-5 pixel), far below one 16-bit step of the blended value.
__device__ __forceinline__ void sw_ray(const float (&hm)[12], float x, float y, float& rx, float& ry, float& rz) {
  rx = __builtin_fmaf(hm[0], x, __builtin_fmaf(hm[1], y, hm[2]));
  ry = __builtin_fmaf(hm[3], x, __builtin_fmaf(hm[4], y, hm[5]));
  rz = __builtin_fmaf(hm[6], x, __builtin_fmaf(hm[7], y, hm[8]));
}
// w: 0 for a corner outside the image (grid_sample padding_mode="zeros": its clamped address is read, times 0) and for a voxel outside the
// image (conv zero padding, inb false); NaN in w[0] for a non-finite projection (the voxel becomes NaN like the reference's).  off: byte
// offsets of the four clamped corner pixels inside the partner feature map (64 bytes per pixel).
__device__ __forceinline__ void sw_corner_weights(float rx, float ry, float rz, float t0, float t1, float t2, float depth, float sx, float sy,
                                                  int W, int H, bool inb, float (&w)[4], unsigned (&off)[4]) {
  const float px = __builtin_fmaf(rx, depth, t0), py = __builtin_fmaf(ry, depth, t1), pz = __builtin_fmaf(rz, depth, t2);
  const float rinv = __builtin_amdgcn_rcpf(pz);
  const float ix = __builtin_fmaf(px * rinv, sx, -0.5f), iy = __builtin_fmaf(py * rinv, sy, -0.5f);
  const bool fin = isfinite(ix) && isfinite(iy);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;                  // v_cvt_i32_f32 saturates: far-away projections stay "outside"
  const float tx = ix - fx, ty = iy - fy;
  const bool xin0 = (unsigned)x0 < (unsigned)W, xin1 = (unsigned)(x0 + 1) < (unsigned)W;
  const bool yin0 = (unsigned)y0 < (unsigned)H, yin1 = (unsigned)(y0 + 1) < (unsigned)H;
  const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0, -1) + 1, W - 1);
  const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0, -1) + 1, H - 1);
  const float ux = 1.f - tx, uy = 1.f - ty;
  w[0] = (inb && xin0 && yin0) ? ux * uy : 0.f;
  w[1] = (inb && xin1 && yin0) ? tx * uy : 0.f;
  w[2] = (inb && xin0 && yin1) ? ux * ty : 0.f;
  w[3] = (inb && xin1 && yin1) ? tx * ty : 0.f;
  if (inb && !fin) w[0] = __builtin_nanf("");
  const unsigned r0 = (unsigned)(yc0 * W), r1 = (unsigned)(yc1 * W);
  off[0] = (r0 + (unsigned)xc0) * 64u;
  off[1] = (r0 + (unsigned)xc1) * 64u;
  off[2] = (r1 + (unsigned)xc0) * 64u;
  off[3] = (r1 + (unsigned)xc1) * 64u;
}

}  // namespace

// The one-tile form.  Register cap of three waves per SIMD (<= 168 VGPRs; the instantiations need 150): with the seven-wave
// workgroups of rounds 1-4 the hardware could start the next workgroup's first waves while this one's consumers finished; an
// eight-wave workgroup (four consumer waves, round 5) leaves room for nothing else on the CU, which is what the persistent
// form below turns into a design (-DSW_OCC=4, two workgroups per CU at 128 registers: 64 bytes of scratch, slower).
// T: storage type of the feature maps and the conv0 weights = the MFMA operand type; TO: storage type of c0.  <f16_t, unsigned short> is the
// bf16 nets' default since round 5 (`final` writes their feature map as f16 for this kernel: adapose.cpp feat_f16()).
template <typename T, typename TO, int BL>
__global__ __launch_bounds__(SW_THREADS, SW_OCC) void conv0_sweep_kernel(const SweepDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char planes[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // XCD-aware tile order: every XCD walks a contiguous run of tiles (whole views) so the partner feature maps its CUs
  // gather from stay in that XCD's L2
  // (with a tile list the launch still covers every tile: workgroups past the list's end leave at once)
  const int nblk = d.tile_list ? d.tile_count[0] : (int)gridDim.x;
  if ((int)blockIdx.x >= nblk) return;
  const int bq = nblk >> 3, br = nblk & 7, xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  // blocks xcd, xcd + 8, ... < nblk: XCD xcd owns (nblk - xcd + 7) / 8 = bq + (xcd < br) of them
  int t = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bidx;
  if (d.tile_list) t = d.tile_list[t];
  const int tw = t % d.ntw; t /= d.ntw;
  const int th = t % d.nth; t /= d.nth;
  const int n = t;
  const int h0 = th * SW_TH, w0 = tw * SW_TW;
  const int D = d.D, H = d.H, W = d.W;
  const int vv = d.v0 + n;

  if (wave < SW_NPW) {
    // ------------------------------------------------------------------ producers
    if (std::is_same<T, f16_t>::value) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);      // MODE.FP16_OVFL = 1 (see blend_chunk<f16_t>)
    if constexpr (BL == 3 && SW_COOP != 0) {
    // ---- cooperative form (packed-f16 blend).  The projection stays one lane per voxel (lane l of producer wave w owns voxel 64 w + l and
    // computes its four corner offsets and weights); the gathers, the blend and the LDS store run per (voxel, 16-byte chunk): in round r
    // lane l handles chunk l & 3 of voxel 64 w + (l & ~3) + r - one of its own quad's four voxels - and takes that voxel's offsets and
    // weights from lane (l & ~3) + r with a DPP quad broadcast (a VALU move; the first version took 16 consecutive voxels per round and
    // fetched with ds_bpermute_b32: 32 LDS round trips per plane made the producers 30 % slower than lane-per-voxel).  A gather instruction
    // touches 16 pixels' 64-byte runs instead of 64 lanes' 16-byte pieces.  Same rolling schedule as the lane-per-voxel form: round r of plane z + 1 is requested right after
    // round r of plane z has been blended out of the same registers; 12 of 16 gathers stay in flight.
    const int ck = lane & 3;
    const int pv = tid;                                  // the projection's voxel
    const bool act = pv < SW_NV;
    const int hh = pv / SW_HW, hw = pv - hh * SW_HW;
    const int gh = h0 - 1 + hh, gw = w0 - 1 + hw;
    const bool inb = act && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
    const int partner = (vv + d.B) % d.V, bb = vv % d.B;
    const float* __restrict__ hm = d.homog + (long long)vv * 12;
    const float* __restrict__ dep = d.depths + (long long)bb * D;
    const unsigned char* __restrict__ srcb = reinterpret_cast<const unsigned char*>(d.feat + (long long)partner * H * W * 32);
    uint4 ref[4];                                        // [round]: chunk ck of the round's voxel
    unsigned dsto[4];                                    // LDS byte offset of that chunk inside a plane slot
    bool actr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cv = wave * 64 + (lane & ~3) + r;
      const int ch_ = cv / SW_HW, cw_ = cv - ch_ * SW_HW;
      const int ggh = h0 - 1 + ch_, ggw = w0 - 1 + cw_;
      actr[r] = cv < SW_NV;
      dsto[r] = (unsigned)(cv * SW_VS + ck * 16);
      ref[r] = make_uint4(0u, 0u, 0u, 0u);               // outside the image: conv zero padding
      if (actr[r] && (unsigned)ggh < (unsigned)H && (unsigned)ggw < (unsigned)W)
        ref[r] = *reinterpret_cast<const uint4*>(d.feat + (((long long)vv * H + ggh) * W + ggw) * 32 + ck * 8);
    }
    float hmv[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) hmv[e] = hm[e];
    float rx, ry, rz;
    sw_ray(hmv, (float)gw, (float)gh, rx, ry, rz);
    const float t0 = hmv[9], t1 = hmv[10], t2 = hmv[11];
    const float sx = (float)W / (float)(W - 1), sy = (float)H / (float)(H - 1);
    const int dbits = __float_as_int(lane < D ? dep[lane] : 1.f);
    // per voxel: four byte offsets of the (clamped) corner pixels and the four weights as f16 pairs (sw_corner_weights)
    auto corners = [&](int z, unsigned (&off)[4], unsigned (&wp)[4]) {
      float w[4];
      sw_corner_weights(rx, ry, rz, t0, t1, t2, __int_as_float(__builtin_amdgcn_readlane(dbits, z)), sx, sy, W, H, inb, w, off);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const Sw16<f16_t>::h2 h = {(f16_t)w[q], (f16_t)w[q]};
        wp[q] = __builtin_bit_cast(unsigned, h);
      }
    };
    u32x4 g[4][4];                                       // [round][corner]
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) g[k][q] = u32x4{0u, 0u, 0u, 0u};
    unsigned wco[4][4];                                  // [round][corner]: weights of the gathers in flight
#define SW_CGATHER(R, Q, OFF) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(g[R][Q]) : "v"(OFF), "s"(srcb) : "memory")
#define SW_CWAIT12(K) asm volatile("s_waitcnt vmcnt(12)" : "+v"(g[K][0]), "+v"(g[K][1]), "+v"(g[K][2]), "+v"(g[K][3]) :: "memory")
    // request round R of the plane whose per-voxel values are (off, wp): every lane fetches its voxel's four offsets and weights
#define SW_CREQ(R, OFFV, WPV)                                                                                         \
    do {                                                                                                              \
      unsigned o_[4];                                                                                                 \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                 \
        o_[q] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(OFFV)[q], (R) * 0x55, 0xf, 0xf, false) | (unsigned)(ck * 16);      \
        wco[R][q] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(WPV)[q], (R) * 0x55, 0xf, 0xf, false);             \
      }                                                                                                               \
      SW_CGATHER(R, 0, o_[0]); SW_CGATHER(R, 1, o_[1]); SW_CGATHER(R, 2, o_[2]); SW_CGATHER(R, 3, o_[3]);             \
    } while (0)
    unsigned noff[4], nwp[4];
    corners(0, noff, nwp);
    SW_CREQ(0, noff, nwp); SW_CREQ(1, noff, nwp); SW_CREQ(2, noff, nwp); SW_CREQ(3, noff, nwp);
    auto blend4 = [&](const uint4& r, const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& e, const unsigned (&w)[4]) {
      const unsigned rr[4] = {r.x, r.y, r.z, r.w}, aa[4] = {a[0], a[1], a[2], a[3]}, bb4[4] = {b[0], b[1], b[2], b[3]};
      const unsigned cc[4] = {c[0], c[1], c[2], c[3]}, ee[4] = {e[0], e[1], e[2], e[3]};
      unsigned o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned t;
        asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(t) : "v"(aa[q]), "v"(w[0]), "v"(rr[q]));
        asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(t) : "v"(bb4[q]), "v"(w[1]));
        asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(t) : "v"(cc[q]), "v"(w[2]));
        asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(t) : "v"(ee[q]), "v"(w[3]));
        o[q] = t;
      }
      return make_uint4(o[0], o[1], o[2], o[3]);
    };
    // (two loops, not `if (z < D)` inside one: the plane loop that holds the counted waits has no branch but its back edge - check_asm_gathers.py)
    for (int z = 0; z < D; ++z) {
      corners(min(z + 1, D - 1), noff, nwp);             // last plane: a harmless re-request keeps the wait counts static
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(ref[k].x), "+v"(ref[k].y), "+v"(ref[k].z), "+v"(ref[k].w));
      unsigned char* dst = planes + (z % SW_NSLOT) * SW_SLOT;
#define SW_CROUND(R)                                                                                                  \
      do {                                                                                                            \
        SW_CWAIT12(R);                                                                                                \
        {                                                                                                             \
          const uint4 o4 = blend4(ref[R], g[R][0], g[R][1], g[R][2], g[R][3], wco[R]);                                \
          if (actr[R]) *reinterpret_cast<uint4*>(dst + dsto[R]) = o4;                                                 \
        }                                                                                                             \
        SW_CREQ(R, noff, nwp);                                                                                        \
      } while (0)
      SW_CROUND(0); SW_CROUND(1); SW_CROUND(2); SW_CROUND(3);
#undef SW_CROUND
      // the plane just written must be visible before the consumers are released; prefetched gathers stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");                     // no LDS store of the next plane may be scheduled above the barrier
    }
    for (int z = D; z < D + SW_LAG; ++z) {               // the consumers run SW_LAG planes behind
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
#undef SW_CGATHER
#undef SW_CWAIT12
#undef SW_CREQ
    } else {
    const int pv = tid;                                  // voxel of the (TH+2) x 18 plane
    const bool act = pv < SW_NV;
    const int hh = pv / SW_HW, hw = pv - hh * SW_HW;
    const int gh = h0 - 1 + hh, gw = w0 - 1 + hw;
    const bool inb = act && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
    const int partner = (vv + d.B) % d.V, bb = vv % d.B;
    const float* __restrict__ hm = d.homog + (long long)vv * 12;
    const float* __restrict__ dep = d.depths + (long long)bb * D;
    const unsigned char* __restrict__ srcb = reinterpret_cast<const unsigned char*>(d.feat + (long long)partner * H * W * 32);
    unsigned char* dst0 = planes + pv * SW_VS;

    uint4 ref[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ref[k] = make_uint4(0u, 0u, 0u, 0u);    // outside the image: conv zero padding
    if (inb) {
      const unsigned short* pr = d.feat + (((long long)vv * H + gh) * W + gw) * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) ref[k] = *reinterpret_cast<const uint4*>(pr + k * 8);
    }

    // projection of this pixel: sw_ray once, sw_corner_weights per plane
    float hmv[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) hmv[e] = hm[e];
    float rx, ry, rz;
    sw_ray(hmv, (float)gw, (float)gh, rx, ry, rz);
    const float t0 = hmv[9], t1 = hmv[10], t2 = hmv[11];
    const float sx = (float)W / (float)(W - 1), sy = (float)H / (float)(H - 1);

    // the D depths of this pose live in one VGPR (lane z holds depth z): a per-plane scalar read instead of a memory load
    const int dbits = __float_as_int(lane < D ? dep[lane] : 1.f);

    auto corners = [&](int z, Corner& c) {
      sw_corner_weights(rx, ry, rz, t0, t1, t2, __int_as_float(__builtin_amdgcn_readlane(dbits, z)), sx, sy, W, H, inb, c.w, c.off);
      if constexpr (BL == 2) {
        c.wp[0] = pack2_bf16(c.w[0], c.w[1]);
        c.wp[1] = pack2_bf16(c.w[2], c.w[3]);
      }
      if constexpr (std::is_same<T, f16_t>::value && BL == 3) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const Sw16<f16_t>::h2 h = {(f16_t)c.w[q], (f16_t)c.w[q]};
          c.wp[q] = __builtin_bit_cast(unsigned, h);
        }
      }
      if constexpr (SW_SKIP && std::is_same<T, unsigned short>::value) {
        // A plane on which all 64 voxels of this wave project outside the partner image (20-30 % of the (wave, plane) pairs at the
        // synthetic camera geometry): the blend is ref + 0, so the reference chunk is stored as it is, and the wave's gathers for that
        // plane all read pixel 0 (one cache line).  The gathers are still ISSUED — the counted waits stay valid for every plane (DESIGN
        // 5c: a counted wait is valid for one issue history) — only their addresses and the blend arithmetic depend on the flag.
        c.skip = __builtin_amdgcn_ballot_w64(c.w[0] != 0.f || c.w[1] != 0.f || c.w[2] != 0.f || c.w[3] != 0.f) == 0ull;      // NaN != 0: a non-finite projection is never skipped
        const unsigned keep = c.skip ? 0u : ~0u;           // wave-uniform mask instead of a branch: no control flow is added to the address path
#pragma unroll
        for (int q = 0; q < 4; ++q) c.off[q] &= keep;
      } else {
        c.skip = false;
      }
    };

    // One register set of 16 x 16 B: chunk k (its 4 corners) of plane z+1 is requested right after chunk k of plane z
    // has been blended out of the same registers, so every gather has a whole step to land and each blend waits for
    // exactly its 4 oldest loads (12 stay in flight).  The gathers are issued from inline asm with tied ("+v") operands
    // and counted by hand: left to hipcc, its wait-count merging at the loop header plus a back-edge register copy
    // ended every step in `s_waitcnt vmcnt(0)` (seen in the ISA), i.e. no prefetch at all.  The asm results must not be
    // touched before gather_wait(): the empty-bodied asm there names them as in/out so every use is ordered after it.
    Corner cur, nxt;
    auto blend = [](const uint4& r, const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& e, const Corner& cn) {
      if (SW_SKIP && std::is_same<T, unsigned short>::value && cn.skip) return r;      // wave-uniform; the gathers and their counted waits around this call are unconditional
      if constexpr (std::is_same<T, f16_t>::value) return blend_chunk_f16<BL == 3>(r, a, b, c, e, cn);
      else return blend_chunk<T, BL>(r, a, b, c, e, cn);
    };
    u32x4 g[4][4];                                       // [chunk][corner]
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) g[k][q] = u32x4{0u, 0u, 0u, 0u};
#define SW_GATHER(K, Q, OFF)                                                                                       \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #K "*16" : "+v"(g[K][Q]) : "v"(OFF), "s"(srcb) : "memory")
#define SW_GATHER4(K, C) do { SW_GATHER(K, 0, (C).off[0]); SW_GATHER(K, 1, (C).off[1]); SW_GATHER(K, 2, (C).off[2]); SW_GATHER(K, 3, (C).off[3]); } while (0)
#define SW_WAIT12(K) asm volatile("s_waitcnt vmcnt(12)" : "+v"(g[K][0]), "+v"(g[K][1]), "+v"(g[K][2]), "+v"(g[K][3]) :: "memory")
    if (act) {
      corners(0, cur);
      SW_GATHER4(0, cur); SW_GATHER4(1, cur); SW_GATHER4(2, cur); SW_GATHER4(3, cur);
    }
    for (int z = 0; z < D + SW_LAG; ++z) {
      if (act && z < D) {
        corners(min(z + 1, D - 1), nxt);                 // last plane: a harmless re-request keeps the wait counts static
        if (sizeof(T) == 2 && !std::is_same<T, unsigned short>::value) {
          // f16_t: keep the reference feature packed (hipcc otherwise hoists its 32 conversions out of the plane loop:
          // 170 VGPRs instead of 164)
#pragma unroll
          for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(ref[k].x), "+v"(ref[k].y), "+v"(ref[k].z), "+v"(ref[k].w));
        }
        unsigned char* dst = dst0 + (z % SW_NSLOT) * SW_SLOT;
        SW_WAIT12(0);
        *reinterpret_cast<uint4*>(dst) = blend(ref[0], g[0][0], g[0][1], g[0][2], g[0][3], cur);
        SW_GATHER4(0, nxt);
        SW_WAIT12(1);
        *reinterpret_cast<uint4*>(dst + 16) = blend(ref[1], g[1][0], g[1][1], g[1][2], g[1][3], cur);
        SW_GATHER4(1, nxt);
        SW_WAIT12(2);
        *reinterpret_cast<uint4*>(dst + 32) = blend(ref[2], g[2][0], g[2][1], g[2][2], g[2][3], cur);
        SW_GATHER4(2, nxt);
        SW_WAIT12(3);
        *reinterpret_cast<uint4*>(dst + 48) = blend(ref[3], g[3][0], g[3][1], g[3][2], g[3][3], cur);
        SW_GATHER4(3, nxt);
        cur = nxt;
      }
      // the plane just written must be visible before the consumers are released; prefetched gathers stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");                     // no LDS store of the next plane may be scheduled above the barrier
    }
    }
#undef SW_GATHER
#undef SW_GATHER4
#undef SW_WAIT12
  } else {
    // ------------------------------------------------------------------ consumers
    const int cw = wave - SW_NPW;
    const int lr = lane & 15, lg = lane >> 4;
    uint4 A01[9], A2[9];
    {
      const uint4* wq = reinterpret_cast<const uint4*>(d.wgt);
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        A01[s] = wq[(s * 16 + lr) * 4 + lg];
        A2[s] = wq[((9 + s) * 16 + lr) * 4 + lg];
      }
    }
    // after the lane-half swap a lane holds: fragment (lg < 2 ? first : second of the pair), voxel lr, channels (lg&1)*4..+3
    const int ch = (lg & 1) * 4;
    // the folded BatchNorm shift is the initial value of the accumulator rows that become outputs (rows 8-15 = lanes 32-63 before the lane-half
    // swap: lane group lg holds rows 4 lg .. 4 lg + 3); rows 0-7 start at 0
    f32x4 binit;
#pragma unroll
    for (int r = 0; r < 4; ++r) binit[r] = lg >= 2 ? d.bias[(lg & 1) * 4 + r] : 0.f;
    const int ow = w0 + lr;
    int oh[SW_NPAIR];
    bool ook[SW_NPAIR];
#pragma unroll
    for (int pr = 0; pr < SW_NPAIR; ++pr) {
      const bool self = 2 * pr + 1 >= SW_CR_;                    // odd row count: the last fragment pairs with itself, lanes lg < 2 store it
      oh[pr] = h0 + cw * SW_CR_ + pr * 2 + (self ? 0 : (lg >> 1));
      ook[pr] = oh[pr] < H && ow < W && !(self && lg >= 2);
    }
    const int boff = ((cw * SW_CR_) * SW_HW + lr) * SW_VS + lg * 16;     // fragment 0, tap (0,0)

    f32x4 Xp[SW_CR_], Lp[SW_NPAIR];
#pragma unroll
    for (int f = 0; f < SW_CR_; ++f) Xp[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < SW_NPAIR; ++pr) Lp[pr] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto emit = [&](int o) {           // out plane o from Xp (= X[o]) and Lp (= rows 0-7 of X[o-1]); leaves Lp = rows 0-7 of X[o]
#pragma unroll
      for (int pr = 0; pr < SW_NPAIR; ++pr) {
        f32x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f1 = 2 * pr + 1 < SW_CR_ ? 2 * pr + 1 : 2 * pr;
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Xp[2 * pr][r]), __float_as_uint(Xp[f1][r]), false, false);
          lo[r] = __uint_as_float(sw[0]);      // lanes 0-31: rows 0-7 of frag 2pr, lanes 32-63: rows 0-7 of frag 2pr+1
          hi[r] = __uint_as_float(sw[1]);      // rows 8-15 likewise
        }
        if (o >= 0 && ook[pr]) {
          float v[4];
          TO* const dst = reinterpret_cast<TO*>(d.out) + ((((long long)n * D + o) * H + oh[pr]) * W + ow) * 8 + ch;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = hi[r] + Lp[pr][r];
            v[r] = v[r] < 0.f ? 0.f : v[r];                        // conv0 always has its ReLU (the launcher checks); NaN propagates, like torch.relu
          }
          store4(dst, v);
        }
        Lp[pr] = lo;
      }
    };

    // operand fragments of a plane: b[r][c] = voxels (row cw * SW_CR + r, columns c .. c + 15) of the 14 x 18 input plane, r < SW_CR + 2
    for (int z = 0; z < D + SW_LAG; ++z) {
      if (z >= SW_LAG) {
        const int p = z - SW_LAG;
        const unsigned char* slot = planes + (p % SW_NSLOT) * SW_SLOT + boff;
        f32x4 Xn[SW_CR_];
#pragma unroll
        for (int f = 0; f < SW_CR_; ++f) Xn[f] = binit;
        uint4 b[SW_CR_ + 2][3];
#pragma unroll
        for (int r = 0; r < SW_CR_ + 2; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            b[r][c] = *reinterpret_cast<const uint4*>(slot + (r * SW_HW + c) * SW_VS);
          }
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
          for (int f = 0; f < SW_CR_; ++f) {
            Xn[f] = Sw16<T>::mma(A01[tp], b[f + tp / 3][tp % 3], Xn[f]);
            Xp[f] = Sw16<T>::mma(A2[tp], b[f + tp / 3][tp % 3], Xp[f]);
          }
        }
        emit(p - 1);                   // X[p-1] is complete once plane p has contributed its kd=2 taps
#pragma unroll
        for (int f = 0; f < SW_CR_; ++f) Xp[f] = Xn[f];
      }
      // The barrier builtin alone does not stop hipcc from hoisting the next plane's first ds_reads above it (seen in the
      // ISA: "ds_read, ds_read, s_barrier"): those reads raced with the producers still writing that slot.  The empty asm
      // with a memory clobber pins every LDS access to its side of the barrier.  The lgkmcnt(0) covers the other direction:
      // MFMAs are not memory operations, so hipcc may sink a plane's last ds_read/MFMA pairs below the barrier (seen in
      // the peeled first iteration of an experimental build) — the read would then still be queued when the producers
      // start to overwrite the slot.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    emit(D - 1);                       // plane D is zero padding: X[D-1] is already complete
  }
}

// ---------------------------------------------------------------------------------------------------------------- persistent form
// Round 5.  Timers in the one-tile-per-workgroup kernel above (-DSW_ABL=256) put a workgroup's life at 60-65 k cycles of which the 25
// steady-state plane iterations are 42 k: the rest is the pipeline's fill - the producers' chain of dependent loads (tile list -> homography,
// depths, reference features -> first corner gathers) in front of plane 0, the consumers' 18 weight loads, the launch of the next
// workgroup once this one's waves have retired.  Here ONE workgroup per CU walks its tiles and the two roles never drain: the producers go
// from plane D - 1 of a tile straight to plane 0 of the next (whose reference features, depths and homography were requested a whole tile
// earlier), the consumers follow one plane behind, keep their weights, and spend one extra output epilogue per tile instead of one
// iteration.  f16 feature maps + packed-f16 blend + cooperative quad gathers only (what the bf16 nets run); TO = storage type of c0.
// (Round 5 also measured the consumers TWO planes behind, reading all 15 operand fragments of the next plane during the current plane's epilogue -
// no LDS round trip between the barrier and the first MFMA: 12.9-13.0 ms dense against 12.5.  A denser MFMA phase takes from the producer on the
// same SIMD what it gives the consumer (fp32 arithmetic of another wave does not issue under MFMAs: tools/micro/mfma_valu_coissue.hip).  The
// `SWP_PRE` switch left the source in round 6.)
constexpr int SWP_LAG = 1;                         // the consumers run one plane behind the producers
constexpr int SWP_NSLOT = 4;                       // 80.6 KB: also keeps a second workgroup off the CU (one workgroup per CU by construction)
constexpr int SWP_LDS = SWP_NSLOT * SW_SLOT;

// MIX (fp16 nets, round 6): the blend accumulates in fp32 with v_fma_mix_f32 on fp32 corner weights and converts once per dword (the
// arithmetic of blend_chunk_f16<false>: c0 within 1e-4 of the halo-tile conv0, where the packed form measures 4.7e-4 and clamps every
// partial sum) - 10 fp32-class instructions per dword instead of 4 packed ones; the persistent walk, the quad gathers and the consumers
// are the same.
template <typename TO, bool MIX>
__global__ __launch_bounds__(SW_THREADS, 2) void conv0_sweep_persistent_kernel(const SweepDesc d) {
  typedef f16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char planes[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = d.D, H = d.H, W = d.W;
  const int nblk = d.tile_list ? d.tile_count[0] : d.N * d.nth * d.ntw;
  // virtual block v = blockIdx.x + i * gridDim.x (the grid is a multiple of 8, so v % 8 is this workgroup's XCD for every i): the tile order
  // of the one-tile kernel - every XCD walks a contiguous run of tiles, its CUs side by side on neighbouring tiles of the same views
  const int bq = nblk >> 3, br = nblk & 7, xcd = blockIdx.x & 7;
  const int xbase = xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq, xcnt = bq + (xcd < br);
  const int nt = ((int)(blockIdx.x >> 3) < xcnt) ? (xcnt - (int)(blockIdx.x >> 3) + (int)(gridDim.x >> 3) - 1) / (int)(gridDim.x >> 3) : 0;
  if (nt <= 0) return;
  auto tile_index = [&](int i) -> int {            // i-th tile of this workgroup (a load when the launch walks a list: callers prefetch)
    const int t = xbase + (int)(blockIdx.x >> 3) + i * (int)(gridDim.x >> 3);
    return __builtin_amdgcn_readfirstlane(d.tile_list ? d.tile_list[t] : t);
  };

  if (wave < SW_NPW) {
    // ------------------------------------------------------------------ producers (see the cooperative form of conv0_sweep_kernel)
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);         // MODE.FP16_OVFL = 1
    const int ck = lane & 3;
    const int pv = tid;
    const bool act = pv < SW_NV;
    const int hh = pv / SW_HW, hw = pv - hh * SW_HW;
    unsigned dsto[4];
    bool actr[4];
    int rhh[4], rhw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cv = wave * 64 + (lane & ~3) + r;
      rhh[r] = cv / SW_HW; rhw[r] = cv - rhh[r] * SW_HW;
      actr[r] = cv < SW_NV;
      dsto[r] = (unsigned)(cv * SW_VS + ck * 16);
    }
    const float sx = (float)W / (float)(W - 1), sy = (float)H / (float)(H - 1);
    // what the requests (projection + gathers) of a tile need, per lane = per voxel of the projection
    struct TileP { int inb; float rx, ry, rz, t0, t1, t2; int dbits; const unsigned char* srcb; };
    struct TileRaw { float hm[12]; int dbits; int gh, gw; const unsigned char* srcb; };      // loads in flight: nothing derived yet
    auto tile_request = [&](int t, TileRaw& R, uint4 (&rf)[4]) {      // issue a tile's loads; nothing here waits for them
      const int tw = t % d.ntw; t /= d.ntw;
      const int th = t % d.nth; t /= d.nth;
      const int n = t, h0 = th * SW_TH, w0 = tw * SW_TW, vv = d.v0 + n;
      const int partner = (vv + d.B) % d.V, bb = vv % d.B;
      const float* __restrict__ hm = d.homog + (long long)vv * 12;
#pragma unroll
      for (int e = 0; e < 12; ++e) R.hm[e] = hm[e];
      R.dbits = __float_as_int(lane < D ? d.depths[(long long)bb * D + lane] : 1.f);
      R.gh = h0 - 1 + hh; R.gw = w0 - 1 + hw;
      R.srcb = reinterpret_cast<const unsigned char*>(d.feat + (long long)partner * H * W * 32);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ggh = h0 - 1 + rhh[r], ggw = w0 - 1 + rhw[r];
        rf[r] = make_uint4(0u, 0u, 0u, 0u);              // outside the image: conv zero padding
        if (actr[r] && (unsigned)ggh < (unsigned)H && (unsigned)ggw < (unsigned)W)
          rf[r] = *reinterpret_cast<const uint4*>(d.feat + (((long long)vv * H + ggh) * W + ggw) * 32 + ck * 8);
      }
    };
    auto tile_finish = [&](const TileRaw& R, TileP& P) {
      P.inb = act && (unsigned)R.gh < (unsigned)H && (unsigned)R.gw < (unsigned)W;
      sw_ray(R.hm, (float)R.gw, (float)R.gh, P.rx, P.ry, P.rz);
      P.t0 = R.hm[9]; P.t1 = R.hm[10]; P.t2 = R.hm[11];
      P.dbits = R.dbits; P.srcb = R.srcb;
    };
    auto corners = [&](int z, const TileP& P, unsigned (&off)[4], unsigned (&wp)[4]) {
      float w[4];
      sw_corner_weights(P.rx, P.ry, P.rz, P.t0, P.t1, P.t2, __int_as_float(__builtin_amdgcn_readlane(P.dbits, z)), sx, sy, W, H, P.inb != 0, w, off);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (MIX) wp[q] = __float_as_uint(w[q]);      // fp32 weights for the v_fma_mix blend
        else {
          const Sw16<f16_t>::h2 h = {(f16_t)w[q], (f16_t)w[q]};
          wp[q] = __builtin_bit_cast(unsigned, h);
        }
      }
    };
    u32x4 g[4][4];                                       // [round][corner]
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) g[k][q] = u32x4{0u, 0u, 0u, 0u};
    unsigned wco[4][4];                                  // [round][corner]: weights of the gathers in flight
#define SWP_GATHER(R, Q, OFF, BASE) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(g[R][Q]) : "v"(OFF), "s"(BASE) : "memory")
#define SWP_WAIT12(K) asm volatile("s_waitcnt vmcnt(12)" : "+v"(g[K][0]), "+v"(g[K][1]), "+v"(g[K][2]), "+v"(g[K][3]) :: "memory")
#define SWP_REQ(R, OFFV, WPV, BASE)                                                                                   \
    do {                                                                                                              \
      unsigned o_[4];                                                                                                 \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                 \
        o_[q] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(OFFV)[q], (R) * 0x55, 0xf, 0xf, false) | (unsigned)(ck * 16);      \
        wco[R][q] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(WPV)[q], (R) * 0x55, 0xf, 0xf, false);             \
      }                                                                                                               \
      SWP_GATHER(R, 0, o_[0], BASE); SWP_GATHER(R, 1, o_[1], BASE); SWP_GATHER(R, 2, o_[2], BASE); SWP_GATHER(R, 3, o_[3], BASE);  \
    } while (0)
    auto blend4 = [&](const uint4& r, const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& e, const unsigned (&w)[4]) {
      const unsigned rr[4] = {r.x, r.y, r.z, r.w}, aa[4] = {a[0], a[1], a[2], a[3]}, bb4[4] = {b[0], b[1], b[2], b[3]};
      const unsigned cc[4] = {c[0], c[1], c[2], c[3]}, ee[4] = {e[0], e[1], e[2], e[3]};
      unsigned o[4];
      if constexpr (MIX) {
        // the sum of blend_chunk_f16<false>, same order: ((a w0 + b w1) + c w2) + e w3, then + ref; MODE.FP16_OVFL saturates the conversion
#define SWP_MIX(HI, D, H, W, C) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[" #HI ",0,0] op_sel_hi:[1,0,0]" : "=v"(D) : "v"(H), "v"(W), "v"(C))
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float lo, hi;
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(aa[q]), "v"(w[0]));
          asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(aa[q]), "v"(w[0]));
          SWP_MIX(0, lo, bb4[q], w[1], lo); SWP_MIX(1, hi, bb4[q], w[1], hi);
          SWP_MIX(0, lo, cc[q], w[2], lo); SWP_MIX(1, hi, cc[q], w[2], hi);
          SWP_MIX(0, lo, ee[q], w[3], lo); SWP_MIX(1, hi, ee[q], w[3], hi);
          asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(rr[q]), "v"(lo));
          asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(rr[q]), "v"(hi));
          const Sw16<f16_t>::h2 h = {(f16_t)lo, (f16_t)hi};
          o[q] = __builtin_bit_cast(unsigned, h);
        }
#undef SWP_MIX
      } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned t;
        asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(t) : "v"(aa[q]), "v"(w[0]), "v"(rr[q]));
        asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(t) : "v"(bb4[q]), "v"(w[1]));
        asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(t) : "v"(cc[q]), "v"(w[2]));
        asm("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(t) : "v"(ee[q]), "v"(w[3]));
        o[q] = t;
      }
      }
      return make_uint4(o[0], o[1], o[2], o[3]);
    };
    // one plane: blend the four rounds of the plane whose gathers are in flight (reference features `ref`), write them into ring slot
    // `slot`, and request (projection at depth index ZREQ with the parameters PREQ) the plane that follows it in the stream
#define SWP_PLANE(PREQ, ZREQ)                                                                                         \
    do {                                                                                                              \
      corners((ZREQ), (PREQ), noff, nwp);                                                                             \
      unsigned char* dst = planes + slot * SW_SLOT;                                                                   \
      SWP_ROUND(0, (PREQ).srcb); SWP_ROUND(1, (PREQ).srcb); SWP_ROUND(2, (PREQ).srcb); SWP_ROUND(3, (PREQ).srcb);     \
      slot = (slot + 1) & (SWP_NSLOT - 1);                                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the plane is visible before the consumers are released */ \
      __builtin_amdgcn_s_barrier();                                                                                   \
      asm volatile("" ::: "memory");                                                                                  \
    } while (0)
#define SWP_ROUND(R, BASE)                                                                                            \
    do {                                                                                                              \
      SWP_WAIT12(R);                                                                                                  \
      const uint4 o4 = blend4(ref[R], g[R][0], g[R][1], g[R][2], g[R][3], wco[R]);                                    \
      if (actr[R]) *reinterpret_cast<uint4*>(dst + dsto[R]) = o4;                                                     \
      SWP_REQ(R, noff, nwp, BASE);                                                                                    \
    } while (0)
    TileRaw raw;
    TileP cur, nxt;
    uint4 ref[4], nref[4];
    unsigned noff[4], nwp[4];
    int slot = 0;
    int t1 = nt > 1 ? tile_index(1) : 0;                 // the next tile's index, fetched a tile ahead
    tile_request(tile_index(0), raw, ref);
    tile_finish(raw, cur);
    corners(0, cur, noff, nwp);
    SWP_REQ(0, noff, nwp, cur.srcb); SWP_REQ(1, noff, nwp, cur.srcb); SWP_REQ(2, noff, nwp, cur.srcb); SWP_REQ(3, noff, nwp, cur.srcb);
    for (int ti = 0; ti < nt; ++ti) {
      const bool has_next = ti + 1 < nt;
      // the next tile's loads travel while this tile's planes are produced; their first use is behind the plane loop
      if (has_next) tile_request(t1, raw, nref);
      const int t2 = ti + 2 < nt ? tile_index(ti + 2) : 0;
      for (int z = 0; z + 1 < D; ++z) {
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(ref[k].x), "+v"(ref[k].y), "+v"(ref[k].z), "+v"(ref[k].w));
        SWP_PLANE(cur, z + 1);
      }
      // last plane of the tile: what is requested behind it is plane 0 of the next tile (none left: a harmless re-request of this
      // plane keeps the counted waits valid)
      if (has_next) tile_finish(raw, nxt); else nxt = cur;
      SWP_PLANE(nxt, has_next ? 0 : D - 1);
      if (has_next) {
        cur = nxt;
#pragma unroll
        for (int r = 0; r < 4; ++r) ref[r] = nref[r];
      }
      t1 = t2;
    }
    for (int k = 0; k < SWP_LAG; ++k) {                  // the consumers run SWP_LAG planes behind
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
#undef SWP_GATHER
#undef SWP_WAIT12
#undef SWP_REQ
#undef SWP_PLANE
#undef SWP_ROUND
  } else {
    // ------------------------------------------------------------------ consumers (the compiler-scheduled plane body of conv0_sweep_kernel)
    const int cw = wave - SW_NPW;
    const int lr = lane & 15, lg = lane >> 4;
    uint4 A01[9], A2[9];
    {
      const uint4* wq = reinterpret_cast<const uint4*>(d.wgt);
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        A01[s] = wq[(s * 16 + lr) * 4 + lg];
        A2[s] = wq[((9 + s) * 16 + lr) * 4 + lg];
      }
    }
    const int ch = (lg & 1) * 4;
    f32x4 binit;
#pragma unroll
    for (int r = 0; r < 4; ++r) binit[r] = lg >= 2 ? d.bias[(lg & 1) * 4 + r] : 0.f;
    const int boff = ((cw * SW_CR_) * SW_HW + lr) * SW_VS + lg * 16;     // fragment 0, tap (0,0)
    __builtin_amdgcn_s_barrier();      // plane 0 of the first tile is complete
    asm volatile("" ::: "memory");
    int slot = 0;                                          // ring slot of the plane to multiply
    int t1 = nt > 1 ? tile_index(1) : 0;
    int tcur = tile_index(0);
    for (int ti = 0; ti < nt; ++ti) {
      const int t2 = ti + 2 < nt ? tile_index(ti + 2) : 0;
      int t = tcur;
      const int tw = t % d.ntw; t /= d.ntw;
      const int th = t % d.nth; t /= d.nth;
      const int n = t, h0 = th * SW_TH, w0 = tw * SW_TW;
      const int ow = w0 + lr;
      int oh[SW_NPAIR];
      bool ook[SW_NPAIR];
#pragma unroll
      for (int pr = 0; pr < SW_NPAIR; ++pr) {
        const bool self = 2 * pr + 1 >= SW_CR_;
        oh[pr] = h0 + cw * SW_CR_ + pr * 2 + (self ? 0 : (lg >> 1));
        ook[pr] = oh[pr] < H && ow < W && !(self && lg >= 2);
      }
      f32x4 Xp[SW_CR_], Lp[SW_NPAIR];
#pragma unroll
      for (int f = 0; f < SW_CR_; ++f) Xp[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pr = 0; pr < SW_NPAIR; ++pr) Lp[pr] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto emit = [&](int o) {         // out plane o from Xp (= X[o]) and Lp (= rows 0-7 of X[o-1]); leaves Lp = rows 0-7 of X[o]
#pragma unroll
        for (int pr = 0; pr < SW_NPAIR; ++pr) {
          f32x4 hi, lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f1 = 2 * pr + 1 < SW_CR_ ? 2 * pr + 1 : 2 * pr;
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Xp[2 * pr][r]), __float_as_uint(Xp[f1][r]), false, false);
            lo[r] = __uint_as_float(sw[0]);
            hi[r] = __uint_as_float(sw[1]);
          }
          if (o >= 0 && ook[pr]) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = hi[r] + Lp[pr][r];
              v[r] = v[r] < 0.f ? 0.f : v[r];            // NaN propagates, like torch.relu
            }
            store4(reinterpret_cast<TO*>(d.out) + ((((long long)n * D + o) * H + oh[pr]) * W + ow) * 8 + ch, v);
          }
          Lp[pr] = lo;
        }
      };
      for (int p = 0; p < D; ++p) {
        const unsigned char* sl = planes + slot * SW_SLOT + boff;
        slot = (slot + 1) & (SWP_NSLOT - 1);
        f32x4 Xn[SW_CR_];
#pragma unroll
        for (int f = 0; f < SW_CR_; ++f) Xn[f] = binit;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
          for (int f = 0; f < SW_CR_; ++f) {
            const uint4 b = *reinterpret_cast<const uint4*>(sl + ((f + tp / 3) * SW_HW + tp % 3) * SW_VS);
            Xn[f] = Sw16<T>::mma(A01[tp], b, Xn[f]);
            Xp[f] = Sw16<T>::mma(A2[tp], b, Xp[f]);
          }
        }
        emit(p - 1);                   // X[p-1] is complete once plane p has contributed its kd=2 taps
#pragma unroll
        for (int f = 0; f < SW_CR_; ++f) Xp[f] = Xn[f];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (both directions: see conv0_sweep_kernel)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      emit(D - 1);                     // plane D is zero padding: X[D-1] is already complete
      tcur = t1; t1 = t2;
    }
  }
}

// Pack conv0 weights [8][32][27] (x folded BN scale) into the consumer's A-fragment order [18][16 rows][4 k-groups][8]:
// steps 0..8 = A01 of in-plane tap t (rows 0-7 kd=0, rows 8-15 kd=1), steps 9..17 = A2 (rows 8-15 kd=2, rows 0-7 zero).
void conv0_sweep_pack(const float* w, const float* scale, std::vector<float>& packed) {
  packed.assign((size_t)18 * 16 * 4 * 8, 0.f);
  for (int s = 0; s < 18; ++s)
    for (int row = 0; row < 16; ++row) {
      const int tpl = s % 9, o = row & 7;
      int kd;
      if (s < 9) kd = row < 8 ? 0 : 1;
      else { if (row < 8) continue; kd = 2; }
      const int widx = kd * 9 + tpl;                   // tpl = kh*3 + kw
      for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 8; ++e) {
          const int c = g * 8 + e;
          packed[(((size_t)s * 16 + row) * 4 + g) * 8 + e] = w[((long long)o * 32 + c) * 27 + widx] * (scale ? scale[o] : 1.f);
        }
    }
}

int launch_conv0_sweep(const Conv3dTileDesc& t, int dtype, hipStream_t s) {
  RGBM_REQUIRE(dtype == BF16 || dtype == F16, "conv0 sweep: 16-bit storage types only");
  RGBM_REQUIRE(!t.feat_f16 || dtype == BF16, "conv0 sweep: feat_f16 is the bf16 nets' option");
  SweepDesc d;
  d.feat = reinterpret_cast<const unsigned short*>(t.feat);
  d.wgt = reinterpret_cast<const unsigned short*>(t.wgt);
  d.bias = t.bias; d.homog = t.homog; d.depths = t.depths;
  d.out = reinterpret_cast<unsigned short*>(t.out);
  d.N = t.N; d.D = t.Di; d.H = t.Hi; d.W = t.Wi; d.v0 = t.v0; d.V = t.V; d.B = t.B; d.relu = t.relu;
  d.nth = (d.H + SW_TH - 1) / SW_TH; d.ntw = (d.W + SW_TW - 1) / SW_TW;
  d.tile_list = t.tile_list; d.tile_count = t.tile_count;
  RGBM_REQUIRE(d.feat && d.wgt && d.bias && d.homog && d.depths && d.out && d.D >= 1 && d.D <= 64 && t.Cout == 8, "conv0 sweep arguments");
  RGBM_REQUIRE(t.relu == 1, "conv0 sweep: the kernel applies conv0's ReLU unconditionally");
  RGBM_REQUIRE((d.tile_list == nullptr) == (d.tile_count == nullptr), "conv0 sweep: tile list and count go together");
  const long long nblk = (long long)d.N * d.nth * d.ntw;
  RGBM_REQUIRE(nblk > 0 && nblk < (1ll << 31), "conv0 sweep grid out of range");
  typedef unsigned short u16;
#define SW_LAUNCH(...)                                                                                             \
  do {                                                                                                             \
    auto kern = conv0_sweep_kernel<__VA_ARGS__>;                                                                   \
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), SW_LDS)) return rc;                       \
    prof_begin_launch(s, t.prof_variant >= 0 ? 14 : -1, t.algo_flops, t.algo_bytes);                               \
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(SW_THREADS), SW_LDS, s, d);                                \
  } while (0)
  // Blend (see blend_chunk / blend_chunk_f16).  bf16 nets whose `final` wrote the feature map as f16 (t.feat_f16, the default): packed f16
  // FMAs (mode 3) - the only form whose instructions fit between another wave's MFMAs (tools/micro/mfma_valu_coissue.hip); three more
  // mantissa bits than the bf16 features had.  fp16 nets keep the fp32-accumulating v_fma_mix form (their c0 stays within 1e-4 of the
  // halo-tile conv0; the packed form measures 4.7e-4 and saturates per partial sum): debug flag 2097152 selects the packed form for them
  // (sweep 15.3 -> 14.6 ms dense).  bf16 feature maps (option sweep_f16 = 0): the plain fp32 blend (mode 0; on the library
  // without packed fp32 instructions the three bf16 modes are level - 44.11 / 44.15 / 44.09 ms per forward - and the dot2 form rounds the
  // bilinear weights to 8 bits); debug flag 4194304 = v_perm + v_dot2_f32_bf16 (the round-4 default), 2097152 = fp32 FMAs from inline asm.
  const bool f21 = (g_debug_flags & (1 << 21)) != 0, f22 = (g_debug_flags & (1 << 22)) != 0;
  if (((dtype == BF16 && t.feat_f16) || dtype == F16) && !(g_debug_flags & (1 << 28))) {
    // the persistent form (debug flag 268435456 = one workgroup per tile, for A/B); fp16 nets (round 6): with their fp32-accumulating blend,
    // debug flag 2097152 = the packed blend for them too
    auto kern = dtype == BF16 ? conv0_sweep_persistent_kernel<u16, false> : f21 ? conv0_sweep_persistent_kernel<f16_t, false>
                                                                                : conv0_sweep_persistent_kernel<f16_t, true>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), SWP_LDS)) return rc;
    int n_cu = 0;
    if (int rc = persistent_grid_cus(&n_cu)) return rc;
    const int grid = nblk < n_cu ? (int)((nblk + 7) / 8 * 8) : n_cu;      // a multiple of 8: XCD = block % 8 for every tile of a workgroup
    prof_begin_launch(s, t.prof_variant >= 0 ? 14 : -1, t.algo_flops, t.algo_bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(SW_THREADS), SWP_LDS, s, d);
  }
  else if (dtype == BF16 && t.feat_f16) SW_LAUNCH(f16_t, u16, 3);
  else if (dtype == BF16 && f22) SW_LAUNCH(u16, u16, 2);
  else if (dtype == BF16 && f21) SW_LAUNCH(u16, u16, 1);
  else if (dtype == BF16) SW_LAUNCH(u16, u16, 0);
  else if (f21) SW_LAUNCH(f16_t, f16_t, 3);
  else SW_LAUNCH(f16_t, f16_t, 0);
#undef SW_LAUNCH
  prof_end_launch(s);
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
