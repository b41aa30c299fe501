// Shared declarations for the rgbmanip_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace rgbm {

// storage type of activations / weights; accumulation is always fp32.  BF16X3: a 4-byte slot per value holding the value
// as a bf16 pair hi + lo (16 significand bits), multiplied as hi*hi + lo*hi + hi*lo on the bf16 matrix pipe — the mode that
// meets the reference's fp32 results to 1e-4 at 3 MFMAs per product instead of the 16x slower fp32 MFMA (see bx3_t below)
enum DType { F32 = 0, BF16 = 1, F16 = 2, BF16X3 = 3 };
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_PRELU = 2, ACT_TANH = 3 };
enum ResMode { RES_NONE = 0, RES_PRE_ACT = 1, RES_POST_ACT = 2 };

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 f16_t;                                     // IEEE half storage (dtype F16); bf16 storage is `unsigned short`
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void set_error(const std::string& s);
int fail(const char* what, const char* file, int line);

#define RGBM_CHECK_HIP(expr)                                                             \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      rgbm::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));               \
      return -2;                                                                         \
    }                                                                                    \
  } while (0)

#define RGBM_REQUIRE(cond, msg)                                                          \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      rgbm::set_error(std::string(msg) + " [" #cond "] at " __FILE__ ":" + std::to_string(__LINE__)); \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize for `fn` on the current device: raised whenever a launch needs more than was set
// for that (device, kernel) so far (a once-per-process flag would leave a second device, or a later wider launch, without it)
int ensure_dynamic_lds(const void* fn, int bytes);
// CU count of the current device, a multiple of 8 (one resident workgroup per CU for the persistent kernels); cached per device
int persistent_grid_cus(int* n_cu);

static inline size_t dtype_size(int dt) { return (dt == F32 || dt == BF16X3) ? 4 : 2; }
static inline int dtype_chunk(int dt) { return 16 / (int)dtype_size(dt); }      // elements per 16-byte chunk
static inline int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }
static inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device helpers -------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  // round to nearest even; NaN stays NaN
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// two floats -> packed bf16x2 with the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN kept)
typedef float rgbm_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 rgbm_b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
  rgbm_f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rgbm_b2));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPerChunk = 4;  // elements per 16 bytes
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<unsigned short> {
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float ld(const unsigned short* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(unsigned short* p, float v) { *p = f32_to_bf16(v); }
};

template <> struct Elem<f16_t> {
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
  __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)(v != v ? v : fminf(fmaxf(v, -65504.f), 65504.f)); }
};
// ---- BF16X3 ("split pair") storage -------------------------------------------------------------------------
// Tensor layout, strides and indexing are those of fp32 (4 bytes per element, 4 elements per 16-byte chunk), but a chunk
// of channels c0..c3 holds   dword0 = hi(c0) | hi(c1) << 16,  dword1 = hi(c2) | hi(c3) << 16,
//                            dword2 = lo(c0) | lo(c1) << 16,  dword3 = lo(c2) | lo(c3) << 16
// with hi = bf16(x) (round to nearest even) and lo = bf16(x - hi): x is kept to 16 significand bits (relative error
// <= 2^-17) in the fp32 exponent range (no saturation, no denormal issue).  The first 8 bytes of a chunk are directly a
// K=4-per-lane bf16 MFMA operand of the hi parts, the second 8 bytes the same for the lo parts; two chunks give the
// K=8-per-lane operands of the 16x16x32 instruction.  A product of two such values on the matrix pipe is
// hi*hi + lo*hi + hi*lo (the lo*lo term, <= 2^-18 relative, is dropped); accumulation is fp32.
struct bx3_t { unsigned raw; };      // one 4-byte slot; never read or written alone except through Elem<bx3_t>

// fp32 -> split chunk and back (4 values)
__device__ __forceinline__ uint4 bx3_split4(float a, float b, float c, float d) {
  uint4 o;
  o.x = pack2_bf16(a, b);
  o.y = pack2_bf16(c, d);
  o.z = pack2_bf16(a - __uint_as_float(o.x << 16), b - __uint_as_float(o.x & 0xffff0000u));     // the differences are exact
  o.w = pack2_bf16(c - __uint_as_float(o.y << 16), d - __uint_as_float(o.y & 0xffff0000u));
  return o;
}
__device__ __forceinline__ void bx3_join4(const uint4& c, float* v) {
  v[0] = __uint_as_float(c.x << 16) + __uint_as_float(c.z << 16);
  v[1] = __uint_as_float(c.x & 0xffff0000u) + __uint_as_float(c.z & 0xffff0000u);
  v[2] = __uint_as_float(c.y << 16) + __uint_as_float(c.w << 16);
  v[3] = __uint_as_float(c.y & 0xffff0000u) + __uint_as_float(c.w & 0xffff0000u);
}
template <> struct Elem<bx3_t> {
  static constexpr int kPerChunk = 4;
  // single element (debug / fetch paths only): halves e of dword pair (e>>1, 2 + (e>>1)) of the element's chunk
  __device__ static __forceinline__ float ld(const bx3_t* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned short* ch = reinterpret_cast<const unsigned short*>(a & ~15ull);
    const int e = (int)((a >> 2) & 3);
    return bf16_to_f32(ch[e]) + bf16_to_f32(ch[4 + e]);
  }
  __device__ static __forceinline__ void st(bx3_t* p, float v) {
    const unsigned long long a = (unsigned long long)p;
    unsigned short* ch = reinterpret_cast<unsigned short*>(a & ~15ull);
    const int e = (int)((a >> 2) & 3);
    const unsigned short h = f32_to_bf16(v);
    ch[e] = h;
    ch[4 + e] = f32_to_bf16(v - bf16_to_f32(h));
  }
};

// fp16 stores saturate at +-65504 instead of overflowing to inf (NaN stays NaN: fminf/fmaxf return the other operand for a NaN
// input, so NaN is routed explicitly)
__device__ __forceinline__ float sat_f16(float v) { return v != v ? v : fminf(fmaxf(v, -65504.f), 65504.f); }

// load/store 4 consecutive elements as floats
__device__ __forceinline__ void load4(const float* p, float v[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const unsigned short* p, float v[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void load4(const f16_t* p, float v[4]) {
  const f16x4 t = *reinterpret_cast<const f16x4*>(p);
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
__device__ __forceinline__ void store4(f16_t* p, const float v[4]) {
  f16x4 t;
  t[0] = (f16_t)sat_f16(v[0]); t[1] = (f16_t)sat_f16(v[1]); t[2] = (f16_t)sat_f16(v[2]); t[3] = (f16_t)sat_f16(v[3]);
  *reinterpret_cast<f16x4*>(p) = t;
}
__device__ __forceinline__ void store4(float* p, const float v[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(unsigned short* p, const float v[4]) {
  uint2 t;
  t.x = pack2_bf16(v[0], v[1]);
  t.y = pack2_bf16(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = t;
}
__device__ __forceinline__ void load4(const bx3_t* p, float v[4]) { bx3_join4(*reinterpret_cast<const uint4*>(p), v); }      // p: chunk-aligned
__device__ __forceinline__ void store4(bx3_t* p, const float v[4]) { *reinterpret_cast<uint4*>(p) = bx3_split4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void unpack_chunk(const uint4& c, float* v, bx3_t /*tag*/) { bx3_join4(c, v); }
__device__ __forceinline__ uint4 pack_chunk(const float* v, bx3_t /*tag*/) { return bx3_split4(v[0], v[1], v[2], v[3]); }
// unpack a 16-byte chunk into floats (4 for f32, 8 for bf16)
__device__ __forceinline__ void unpack_chunk(const uint4& c, float* v, float /*tag*/) {
  v[0] = __uint_as_float(c.x); v[1] = __uint_as_float(c.y); v[2] = __uint_as_float(c.z); v[3] = __uint_as_float(c.w);
}
__device__ __forceinline__ void unpack_chunk(const uint4& c, float* v, unsigned short /*tag*/) {
  v[0] = __uint_as_float(c.x << 16); v[1] = __uint_as_float(c.x & 0xffff0000u);
  v[2] = __uint_as_float(c.y << 16); v[3] = __uint_as_float(c.y & 0xffff0000u);
  v[4] = __uint_as_float(c.z << 16); v[5] = __uint_as_float(c.z & 0xffff0000u);
  v[6] = __uint_as_float(c.w << 16); v[7] = __uint_as_float(c.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack_chunk(const uint4& c, float* v, f16_t /*tag*/) {
  const f16x8 h = __builtin_bit_cast(f16x8, c);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
}
__device__ __forceinline__ uint4 pack_chunk(const float* v, f16_t /*tag*/) {
  f16x8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = (f16_t)sat_f16(v[e]);
  return __builtin_bit_cast(uint4, h);
}
__device__ __forceinline__ uint4 pack_chunk(const float* v, float /*tag*/) {
  return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
__device__ __forceinline__ uint4 pack_chunk(const float* v, unsigned short /*tag*/) {
  uint4 c;
  c.x = pack2_bf16(v[0], v[1]);
  c.y = pack2_bf16(v[2], v[3]);
  c.z = pack2_bf16(v[4], v[5]);
  c.w = pack2_bf16(v[6], v[7]);
  return c;
}

// One 16-byte bx3 chunk per lane on each side (4 k values, hi + lo): c += a*b as hi*hi + lo*hi + hi*lo on
// v_mfma_f32_16x16x16_bf16 (small terms first).  Lane group g of the MFMA holds k = 4g..4g+3 = the 4 channels of its chunk,
// for A and B alike.
typedef __attribute__((ext_vector_type(4))) short rgbm_s16x4;
__device__ __forceinline__ f32x4 mma_bx3_k16(const uint4& a, const uint4& b, f32x4 c) {
  const rgbm_s16x4 ah = __builtin_bit_cast(rgbm_s16x4, make_uint2(a.x, a.y)), al = __builtin_bit_cast(rgbm_s16x4, make_uint2(a.z, a.w));
  const rgbm_s16x4 bh = __builtin_bit_cast(rgbm_s16x4, make_uint2(b.x, b.y)), bl = __builtin_bit_cast(rgbm_s16x4, make_uint2(b.z, b.w));
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c, 0, 0, 0);
  return c;
}
// Two chunks per lane and side (8 k values): the full-rate 16x16x32 instruction.  (a0, a1) and (b0, b1) must pair the same
// two chunks of the K tile on both sides.
__device__ __forceinline__ void bx3_pair(const uint4& c0, const uint4& c1, uint4& hi, uint4& lo) {
  hi = make_uint4(c0.x, c0.y, c1.x, c1.y);
  lo = make_uint4(c0.z, c0.w, c1.z, c1.w);
}

// ---- generic implicit-GEMM convolution descriptor (conv_igemm.hip) -------------------
struct ConvDesc {
  const void* in; const void* wgt; void* out; const float* bias; const void* res;
  int N, Di, Hi, Wi, Cin, lcin;           // input tensor NDHWC, Cin power of two
  int Dq, Hq, Wq;                         // enumeration grid of output positions
  int sd, sh, sw, pd, ph, pw;             // input coord = q*s - p + k*dil
  int KD, KH, KW, dild, dilh, dilw;
  int ntaps, KT, Kpad;                    // taps, #K tiles, weight row stride (elements)
  int Cout, ldo;                          // real out channels (mult of 4), out channel stride
  int Do, Ho, Wo;                         // output tensor dims
  int osd, osh, osw, opd, oph, opw;       // output coord = q*os + op
  int act; float slope; int res_mode; int bias_stride;  // bias index = n*bias_stride + ch
  long long M;                            // N*Dq*Hq*Wq
  int n_pix_tiles, n_ch_tiles;
  // persistent kernel only (filled by its launcher): tiles in total, and exact-division magics for Wq, Hq, Dq
  int n_tiles; unsigned fd_m[3]; int fd_s[3];
  int korder;                             // K-tile walk of the request waves: 1 = channel block outer, taps inner; 0 = taps outer
  int buf_ok;                             // request waves may use 32-bit buffer offsets for both operands (set by the launcher)
  const void* ident;                      // conv_igemm_m32_kernel<256 pixels>: 256 x 256 identity in the storage type (residual steps), set by the launcher
  long long m0;                           // first GEMM row of this launch (conv_igemm_m32_kernel: a layer split into a main and a tail launch)
  // optional 1x1 conv fused behind the activation (conv_igemm_ws64_kernel only): y2 = act2(W2 . act(y) + bias2); when set,
  // `out` is not written.  w2: [cout2_pad][kpad2] K-major bf16 rows of 64 input channels.
  const void* w2; const float* bias2; void* out2; int ldo2, cout2, kpad2, act2; float slope2;
  double algo_flops, algo_bytes;          // algorithmic work of this launch (profiling only)
  int out_f32;                            // bf16x3 tensors, generic kernel only: write the result as PLAIN fp32 (same 4-byte slots) instead of split pairs
  // conv_igemm_m32_kernel, 128-pixel tiles, launches of few tiles (small batches): the K loop of a tile is cut into `ksplit` parts, one workgroup
  // each; a part leaves its fp32 accumulators in kscratch, the LAST of a tile's parts to arrive (kcount, per tile and multiply wave) adds the
  // parts in ascending order and runs the epilogue (conv_igemm_m32.inc).  0 / 1 = no split.
  int ksplit; float* kscratch; unsigned* kcount;
};

int launch_conv(const ConvDesc& d, int dtype, hipStream_t s);
// true if launch_conv would run d on the three-role 64-channel kernel (the only one that can fuse a trailing 1x1)
bool conv_ws64_eligible(const ConvDesc& d, int dtype);
// picks the channel-tile size used by launch_conv for Cout (weights must be padded to it)
int conv_ch_tile(int Cout);
int conv_bk(int dtype);  // K-tile in elements

}  // namespace rgbm
